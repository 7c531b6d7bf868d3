// Development probe (GPU box): the copy engine bringing finished bytes into a page-locked RING that host threads read from
// (deflate_bounce.inc's way out).  Measured 20 GB/s there for 4 MiB transfers against 56 GB/s for one 36 MiB transfer into a
// buffer nobody reads (d2h_engine.hip): which of the differences is it?
//   (a) transfer size, into fresh page-locked memory nobody has touched since
//   (b) the same after CPU threads have READ the destination (its lines sit in their caches)
//   (c) the same while CPU threads spin on the completion signal / on a mutex
//   (d) destination flushed from the caches (clflushopt) after the CPU read it
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_ring tools/probes/d2h_ring.hip -lhsa-runtime64 -lpthread && /tmp/d2h_ring
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static hsa_agent_t g_gpu, g_cpu;
static bool agents(const void* dev, const void* host) {
    hsa_amd_pointer_info_t pd, ph;
    memset(&pd, 0, sizeof pd);
    memset(&ph, 0, sizeof ph);
    pd.size = sizeof pd;
    ph.size = sizeof ph;
    if (hsa_amd_pointer_info(const_cast<void*>(dev), &pd, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (hsa_amd_pointer_info(const_cast<void*>(host), &ph, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    g_gpu = pd.agentOwner;
    g_cpu = ph.agentOwner;
    return true;
}

static uint64_t sum_bytes(const uint8_t* p, size_t n) {
    uint64_t s = 0;
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    for (size_t i = 0; i < n / 8; i++) s += q[i];
    return s;
}
static void flush(const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i += 64) _mm_clflushopt(const_cast<uint8_t*>(p + i));
    _mm_sfence();
}

int main() {
    const size_t RING = 64u << 20;
    CK(hipSetDevice(0));
    uint8_t *d_src, *ring;
    CK(hipMalloc(&d_src, RING));
    CK(hipMemset(d_src, 0x5a, RING));
    CK(hipHostMalloc(&ring, RING, hipHostMallocDefault));
    memset(ring, 0, RING);
    CK(hipDeviceSynchronize());
    if (!agents(d_src, ring)) {
        printf("no agents\n");
        return 1;
    }
    hsa_signal_t sig[64];
    for (auto& s : sig) hsa_signal_create(0, 0, nullptr, &s);
    volatile uint64_t sink = 0;
    auto transfer_set = [&](const char* what, size_t sub, size_t total, int mode, int readers) {
        // mode 0: nothing; 1: readers read the landed bytes (as the bounce threads do); 2: ... and flush them afterwards
        double best = 1e30, worst = 0;
        for (int rep = 0; rep < 6; rep++) {
            const size_t n = total / sub;
            std::atomic<int> go{0};
            std::vector<std::thread> th;
            std::atomic<size_t> next{0};
            for (int r = 0; r < readers; r++)
                th.emplace_back([&] {
                    while (!go.load()) _mm_pause();
                    for (;;) {
                        const size_t k = next.fetch_add(1);
                        if (k >= total / (1u << 20)) break;
                        const size_t off = k << 20, rec = off / sub;
                        hsa_signal_wait_scacquire(sig[rec % 64], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
                        if (mode >= 1) sink = sink + sum_bytes(ring + off, 1u << 20);
                        if (mode >= 2) flush(ring + off, 1u << 20);
                    }
                });
            for (size_t k = 0; k < n; k++) hsa_signal_store_relaxed(sig[k % 64], 1);
            const double t0 = now_us();
            go.store(1);
            for (size_t k = 0; k < n; k++)
                if (hsa_amd_memory_async_copy(ring + k * sub, g_cpu, d_src + k * sub, g_gpu, sub, 0, nullptr, sig[k % 64]) != HSA_STATUS_SUCCESS) printf("copy failed\n");
            for (size_t k = 0; k < n; k++) hsa_signal_wait_scacquire(sig[k % 64], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            const double t1 = now_us();
            for (auto& t : th) t.join();
            if (rep >= 1) {
                if (t1 - t0 < best) best = t1 - t0;
                if (t1 - t0 > worst) worst = t1 - t0;
            }
        }
        printf("%-58s %5.1f MiB in %4.1f MiB transfers: landed in %7.1f .. %7.1f us = %5.1f GB/s (best)\n", what, total / 1048576.0, sub / 1048576.0, best, worst,
               total / best / 1e3);
    };
    for (size_t sub : {1u << 20, 4u << 20, 16u << 20, 36u << 20}) transfer_set("(a) nobody reads the destination", sub, sub >= (16u << 20) ? sub * (sub == (36u << 20) ? 1 : 2) : 32u << 20, 0, 0);
    for (size_t sub : {1u << 20, 4u << 20, 16u << 20}) transfer_set("(b) 8 threads read every MiB as it lands", sub, 32u << 20, 1, 8);
    for (size_t sub : {4u << 20}) transfer_set("(b1) 1 thread reads every MiB as it lands", sub, 32u << 20, 1, 1);
    for (size_t sub : {4u << 20}) transfer_set("(c) 8 threads only wait for the signals", sub, 32u << 20, 0, 8);
    for (size_t sub : {1u << 20, 4u << 20}) transfer_set("(d) 8 threads read and flush (clflushopt) every MiB", sub, 32u << 20, 2, 8);
    for (size_t sub : {4u << 20}) transfer_set("(a') nobody reads, after the flushes", sub, 32u << 20, 0, 0);
    // (e) one after the other: issue a transfer only when the one before has landed (no queue on the engine)
    for (size_t sub : {1u << 20, 4u << 20}) {
        double best = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            const size_t n = (32u << 20) / sub;
            const double t0 = now_us();
            for (size_t k = 0; k < n; k++) {
                hsa_signal_store_relaxed(sig[0], 1);
                hsa_amd_memory_async_copy(ring + k * sub, g_cpu, d_src + k * sub, g_gpu, sub, 0, nullptr, sig[0]);
                hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            }
            const double t1 = now_us();
            if (t1 - t0 < best) best = t1 - t0;
        }
        printf("(e) one at a time, %4.1f MiB transfers: %7.1f us per transfer = %5.1f GB/s\n", sub / 1048576.0, best / ((32u << 20) / sub), (32u << 20) / best / 1e3);
    }
    return 0;
}
