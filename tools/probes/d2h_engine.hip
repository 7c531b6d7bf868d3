// Development probe (GPU box): who carries a 36 MB device-to-host copy into page-locked memory, and what does it cost a
// compute kernel that holds every CU the way k_match3 does (1024 threads, 131 KB of LDS, 128 vector registers)?
//   (1) hipMemcpyAsync, alone and beside the kernel
//   (2) hsa_amd_memory_async_copy (the copy engine, SDMA), alone and beside the kernel
//   (3) a store kernel of G workgroups into the device's view of the page-locked buffer, alone
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_engine tools/probes/d2h_engine.hip -lhsa-runtime64 && /tmp/d2h_engine
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__global__ __launch_bounds__(1024) void busy(uint32_t* out, int iters) {
    __shared__ uint32_t lds[131072 / 4];
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9e3779b9u, c = a + 77u, d = b * 3u;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; i++) {
        a = a * 1664525u + b;
        b = (b ^ (a >> 7)) + c;
        c = c * 22695477u + d;
        d = (d ^ (c >> 9)) + lds[(a >> 5) & 8191];
    }
    asm volatile("v_mov_b32 v127, %0" ::"v"(a) : "v127");
    if ((a ^ b ^ c ^ d) == 0x12345u) out[blockIdx.x] = a;
}

__global__ __launch_bounds__(256) void store_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

static hsa_agent_t g_gpu, g_cpu;
static int g_have_gpu = 0, g_have_cpu = 0;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) {
        g_gpu = a;
        g_have_gpu = 1;
    }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) {
        g_cpu = a;
        g_have_cpu = 1;
    }
    return HSA_STATUS_SUCCESS;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t N = 36u << 20;
    CK(hipSetDevice(0));
    uint8_t *d_src, *h_dst;
    uint32_t* d_out;
    CK(hipMalloc(&d_src, N));
    CK(hipMemset(d_src, 0x5a, N));
    CK(hipMalloc(&d_out, 1 << 20));
    CK(hipHostMalloc(&h_dst, N, hipHostMallocDefault));
    memset(h_dst, 0, N);
    hipStream_t sk, sc;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t k0, k1, c0, c1;
    CK(hipEventCreate(&k0));
    CK(hipEventCreate(&k1));
    CK(hipEventCreate(&c0));
    CK(hipEventCreate(&c1));
    const int WG = 256 * 12, IT = 9000;
    float ms;
    // warm
    hipLaunchKernelGGL(busy, dim3(256), dim3(1024), 0, sk, d_out, 100);
    CK(hipMemcpyAsync(h_dst, d_src, N, hipMemcpyDeviceToHost, sc));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(k0, sk));
        hipLaunchKernelGGL(busy, dim3(WG), dim3(1024), 0, sk, d_out, IT);
        CK(hipEventRecord(k1, sk));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, k0, k1));
        printf("busy kernel alone: %.3f ms\n", ms);
    }
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(c0, sc));
        CK(hipMemcpyAsync(h_dst, d_src, N, hipMemcpyDeviceToHost, sc));
        CK(hipEventRecord(c1, sc));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, c0, c1));
        printf("hipMemcpyAsync D2H alone: %.3f ms = %.1f GB/s\n", ms, N / ms / 1e6);
    }
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(k0, sk));
        hipLaunchKernelGGL(busy, dim3(WG), dim3(1024), 0, sk, d_out, IT);
        CK(hipEventRecord(k1, sk));
        CK(hipEventRecord(c0, sc));
        CK(hipMemcpyAsync(h_dst, d_src, N, hipMemcpyDeviceToHost, sc));
        CK(hipEventRecord(c1, sc));
        CK(hipDeviceSynchronize());
        float mk;
        CK(hipEventElapsedTime(&mk, k0, k1));
        CK(hipEventElapsedTime(&ms, c0, c1));
        printf("beside each other: busy %.3f ms, hipMemcpyAsync D2H %.3f ms = %.1f GB/s\n", mk, ms, N / ms / 1e6);
    }
    // ---- HSA: the copy engine chosen by the caller ----
    if (hsa_init() != HSA_STATUS_SUCCESS) {
        printf("hsa_init failed\n");
        return 1;
    }
    hsa_iterate_agents(agent_cb, nullptr);
    printf("agents: gpu %d cpu %d\n", g_have_gpu, g_have_cpu);
    uint32_t mask = 0;
    hsa_status_t st = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask);
    printf("copy_engine_status(dst cpu, src gpu): status %d mask 0x%x\n", (int)st, mask);
    uint32_t pref = 0;
    st = hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &pref);
    printf("preferred engines: status %d mask 0x%x\n", (int)st, pref);
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            memset(h_dst, 0, 4096);
            hsa_signal_store_relaxed(sig, 1);
            if (mode == 1) {
                CK(hipEventRecord(k0, sk));
                hipLaunchKernelGGL(busy, dim3(WG), dim3(1024), 0, sk, d_out, IT);
                CK(hipEventRecord(k1, sk));
            }
            const double t0 = now_us();
            st = hsa_amd_memory_async_copy(h_dst, g_cpu, d_src, g_gpu, N, 0, nullptr, sig);
            if (st != HSA_STATUS_SUCCESS) {
                printf("hsa_amd_memory_async_copy failed: %d\n", (int)st);
                break;
            }
            hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            const double t1 = now_us();
            float mk = 0;
            if (mode == 1) {
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&mk, k0, k1));
            }
            printf("hsa_amd_memory_async_copy D2H %s: %.3f ms = %.1f GB/s (first byte %02x)%s", mode ? "beside busy" : "alone",
                   (t1 - t0) / 1e3, N / (t1 - t0) / 1e3, h_dst[0], mode ? "" : "\n");
            if (mode) printf(", busy %.3f ms\n", mk);
        }
    }
    // engines one by one
    for (int eng = 0; eng < 8; eng++) {
        if (!(mask & (1u << eng))) continue;
        hsa_signal_store_relaxed(sig, 1);
        const double t0 = now_us();
        st = hsa_amd_memory_async_copy_on_engine(h_dst, g_cpu, d_src, g_gpu, N, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << eng), false);
        if (st != HSA_STATUS_SUCCESS) {
            printf("engine %d: status %d\n", eng, (int)st);
            continue;
        }
        hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
        const double t1 = now_us();
        printf("engine %d alone: %.3f ms = %.1f GB/s\n", eng, (t1 - t0) / 1e3, N / (t1 - t0) / 1e3);
    }
    // ---- a store kernel into the mapped buffer ----
    uint8_t* h_dev = nullptr;
    CK(hipHostGetDevicePointer((void**)&h_dev, h_dst, 0));
    const int gs[] = {16, 64, 256, 1024, 4096};
    for (int gi = 0; gi < 5; gi++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(c0, sc));
            hipLaunchKernelGGL(store_kernel, dim3(gs[gi]), dim3(256), 0, sc, (const uint4*)d_src, (uint4*)h_dev, N / 16);
            CK(hipEventRecord(c1, sc));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, c0, c1));
            if (rep) printf("store kernel, %4d workgroups: %.3f ms = %.1f GB/s\n", gs[gi], ms, N / ms / 1e6);
        }
    }
    // the store kernel beside the busy kernel (launched first, so that it holds the CUs)
    for (int gi = 1; gi < 4; gi++) {
        CK(hipEventRecord(k0, sk));
        hipLaunchKernelGGL(busy, dim3(WG), dim3(1024), 0, sk, d_out, IT);
        CK(hipEventRecord(k1, sk));
        CK(hipEventRecord(c0, sc));
        hipLaunchKernelGGL(store_kernel, dim3(gs[gi]), dim3(256), 0, sc, (const uint4*)d_src, (uint4*)h_dev, N / 16);
        CK(hipEventRecord(c1, sc));
        CK(hipDeviceSynchronize());
        float mk;
        CK(hipEventElapsedTime(&mk, k0, k1));
        CK(hipEventElapsedTime(&ms, c0, c1));
        printf("beside each other: busy %.3f ms, store kernel %4d workgroups %.3f ms = %.1f GB/s\n", mk, gs[gi], ms, N / ms / 1e6);
    }
    // ---- both directions at once (the host call: later pieces still arriving while the first ones leave) ----
    {
        const size_t NH = 100u << 20;
        uint8_t *d_in2, *h_src;
        CK(hipMalloc(&d_in2, NH));
        CK(hipHostMalloc(&h_src, NH, hipHostMallocDefault));
        memset(h_src, 1, NH);
        hipStream_t sh;
        CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
        hipEvent_t h0, h1;
        CK(hipEventCreate(&h0));
        CK(hipEventCreate(&h1));
        for (int withk = 0; withk < 2; withk++)
            for (int rep = 0; rep < 3; rep++) {
                if (withk) {
                    CK(hipEventRecord(k0, sk));
                    hipLaunchKernelGGL(busy, dim3(WG), dim3(1024), 0, sk, d_out, IT);
                    CK(hipEventRecord(k1, sk));
                }
                CK(hipEventRecord(h0, sh));
                CK(hipMemcpyAsync(d_in2, h_src, NH, hipMemcpyHostToDevice, sh));
                CK(hipEventRecord(h1, sh));
                CK(hipEventRecord(c0, sc));
                for (int piece = 0; piece < 4; piece++)
                    CK(hipMemcpyAsync(h_dst + piece * (N / 4), d_src + piece * (N / 4), N / 4, hipMemcpyDeviceToHost, sc));
                CK(hipEventRecord(c1, sc));
                CK(hipDeviceSynchronize());
                float mh, mk = 0;
                CK(hipEventElapsedTime(&mh, h0, h1));
                CK(hipEventElapsedTime(&ms, c0, c1));
                if (withk) CK(hipEventElapsedTime(&mk, k0, k1));
                printf("both directions%s: H2D 100 MiB %.3f ms = %.1f GB/s, D2H 36 MiB in 4 pieces %.3f ms = %.1f GB/s, busy %.3f ms\n",
                       withk ? " beside busy" : "", mh, NH / mh / 1e6, ms, N / ms / 1e6, mk);
            }
        // H2D engine by engine
        uint32_t hmask = 0, hpref = 0;
        hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &hmask);
        hsa_amd_memory_get_preferred_copy_engine(g_gpu, g_cpu, &hpref);
        printf("H2D engines: mask 0x%x preferred 0x%x\n", hmask, hpref);
        for (int eng = 0; eng < 4; eng++) {
            if (!(hmask & (1u << eng))) continue;
            hsa_signal_store_relaxed(sig, 1);
            const double t0 = now_us();
            st = hsa_amd_memory_async_copy_on_engine(d_in2, g_gpu, h_src, g_cpu, NH, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << eng), false);
            if (st != HSA_STATUS_SUCCESS) {
                printf("H2D engine %d: status %d\n", eng, (int)st);
                continue;
            }
            hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            const double t1 = now_us();
            printf("H2D engine %d alone: %.3f ms = %.1f GB/s\n", eng, (t1 - t0) / 1e3, NH / (t1 - t0) / 1e3);
        }
    }
    return 0;
}
