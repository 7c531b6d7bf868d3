// Development probe (GPU box): does gfx950 return the bytes at an odd LDS address for ds_read_u16 / b32 / b64 / b128,
// or does it force the address down to the natural alignment?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_unaligned tools/probes/lds_unaligned.hip && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s;
    const uint32_t a = base + 16 + threadIdx.x;  // offsets 0..63
    uint32_t u16, b32, b64lo, b64hi, q0, q1, q2, q3;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(u16) : "v"(a) : "memory");
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b32) : "v"(a) : "memory");
    uint64_t b64;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b64) : "v"(a) : "memory");
    b64lo = (uint32_t)b64;
    b64hi = (uint32_t)(b64 >> 32);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 q;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a) : "memory");
    q0 = q.x; q1 = q.y; q2 = q.z; q3 = q.w;
    uint32_t* o = out + 8 * threadIdx.x;
    o[0] = u16; o[1] = b32; o[2] = b64lo; o[3] = b64hi; o[4] = q0; o[5] = q1; o[6] = q2; o[7] = q3;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[64 * 8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok16 = 1, ok32 = 1, ok64 = 1, ok128 = 1;
    for (int t = 0; t < 16; t++) {
        uint32_t a = 16 + t;
        uint32_t e16 = a | ((a + 1) << 8), e32 = e16 | ((a + 2) << 16) | ((a + 3) << 24);
        uint32_t e32b = (a + 4) | ((a + 5) << 8) | ((a + 6) << 16) | ((a + 7) << 24);
        printf("off %2d: u16 %04x (want %04x)  b32 %08x (want %08x)  b64 %08x %08x  b128 %08x %08x %08x %08x\n", t, h[8 * t], e16,
               h[8 * t + 1], e32, h[8 * t + 2], h[8 * t + 3], h[8 * t + 4], h[8 * t + 5], h[8 * t + 6], h[8 * t + 7]);
        ok16 &= h[8 * t] == e16;
        ok32 &= h[8 * t + 1] == e32;
        ok64 &= h[8 * t + 2] == e32 && h[8 * t + 3] == e32b;
        ok128 &= h[8 * t + 4] == e32 && h[8 * t + 5] == e32b;
    }
    printf("unaligned ok: u16 %d b32 %d b64 %d b128 %d\n", ok16, ok32, ok64, ok128);
    return 0;
}
