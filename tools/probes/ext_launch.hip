// development aid (GPU box): does a timing event attached to a kernel's own dispatch (hipExtLaunchKernelGGL start / stop events)
// cost the queue what a hipEventRecord between two kernels does (5.7 us of idle)?  Six dependent 20 us kernels on one stream:
// (a) bare, (b) with hipEventRecord between them, (c) with every launch carrying its own start / stop events.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ext_launch tools/probes/ext_launch.hip && /tmp/ext_launch
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long cycles, unsigned* out) {
    const unsigned long long t0 = wall_clock64();
    while ((unsigned long long)wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0]++;
}
int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    const unsigned long long c20 = (unsigned long long)rate * 20 / 1000;  // 20 us
    hipEvent_t ev[16];
    for (auto& e : ev) hipEventCreate(&e);
    auto wall = [&](int mode) {
        double best = 1e9;
        for (int rep = 0; rep < 20; rep++) {
            hipStreamSynchronize(st);
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 6; k++) {
                if (mode == 2)
                    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ev[2 * k], ev[2 * k + 1], 0, c20, d);
                else
                    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, c20, d);
                if (mode == 1) hipEventRecord(ev[k], st);
            }
            hipStreamSynchronize(st);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        return best;
    };
    const double a = wall(0), b = wall(1), c = wall(2);
    printf("wall clock rate %d kHz; six 20 us kernels: bare %.1f us, hipEventRecord between them %.1f us, start/stop events on the launches %.1f us\n",
           rate, a * 1e3, b * 1e3, c * 1e3);
    float e01 = 0, k0 = 0, k5 = 0, span = 0;
    wall(1);
    hipEventElapsedTime(&e01, ev[0], ev[1]);
    printf("  recorded events: elapsed between the first two %.1f us\n", e01 * 1e3);
    wall(2);
    hipEventElapsedTime(&k0, ev[0], ev[1]);
    hipEventElapsedTime(&k5, ev[10], ev[11]);
    hipEventElapsedTime(&span, ev[0], ev[11]);
    printf("  launch events: kernel 0 %.1f us, kernel 5 %.1f us, start of 0 to stop of 5 %.1f us\n", k0 * 1e3, k5 * 1e3, span * 1e3);
    return 0;
}
