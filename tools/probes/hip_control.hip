// Control for the crash hunt: a HIP program that does nothing of ours -- stream, pinned + device memory, an async
// copy each way, one kernel, events, teardown -- run as often as the CLI example, to tell a crash of the runtime's
// own start-up / exit path from one of the library.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hip_control tools/probes/hip_control.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void k(unsigned* p, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 2654435761u + i;
}
int main() {
    const unsigned n = 1u << 20;
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return 3;
    unsigned *d = nullptr, *h = nullptr;
    if (hipMalloc((void**)&d, n * 4) != hipSuccess || hipHostMalloc((void**)&h, 4096, hipHostMallocDefault) != hipSuccess) return 3;
    std::vector<unsigned> v(n, 7u), w(n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    hipMemcpyAsync(d, v.data(), n * 4, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, st, d, n);
    hipMemcpyAsync(w.data(), d, n * 4, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int ok = w[5] == 7u * 2654435761u + 5 && h[5] == w[5];
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    hipHostFree(h);
    hipStreamDestroy(st);
    return ok ? 0 : 4;
}
