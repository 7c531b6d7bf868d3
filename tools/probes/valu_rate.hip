// Development probe (GPU box): how many cycles does a wave64 vector-ALU instruction hold its SIMD on gfx950 -- four (16 lanes a
// cycle) or two (32)?  Everything about k_match3's bound hangs on it ("17.1 wave-instructions per byte").
// W waves per SIMD run a loop of independent (and, in a second variant, dependent) integer instructions of the kinds the walk
// is made of; the clock is s_memtime of the wave itself.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(uint64_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 77u, d = a ^ 0x55u, e = a + 5u, f = a ^ 9u, g = a + 11u, h = a ^ 13u;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // eight independent chains of v_add_u32
            asm volatile(REP8("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                              "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                         : "v"(seed));
        } else if (KIND == 1) {  // one dependent chain
            asm volatile(REP8("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
                              "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t")
                         : "+v"(a)
                         : "v"(seed));
        } else if (KIND == 2) {  // the mix of a service: xor, ffbl, perm, alignbyte, cndmask, min (independent pairs)
            asm volatile(REP8("v_xor_b32 %0, %0, %8\n\tv_ffbl_b32 %1, %0\n\tv_perm_b32 %2, %2, %8, %3\n\tv_alignbyte_b32 %3, %3, %8, %4\n\t"
                              "v_cndmask_b32 %4, %4, %8, vcc\n\tv_min_u32 %5, %5, %8\n\tv_or_b32 %6, %6, %8\n\tv_lshrrev_b32 %7, 3, %7\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                         : "v"(seed)
                         : "vcc");
        } else {  // v_cmpx chain as in the step block (writes exec; restored)
            asm volatile("s_mov_b64 s[10:11], exec\n\t" REP8("v_cmpx_ne_u32 vcc, %0, %1\n\tv_cmpx_ne_u32 vcc, %0, %1\n\tv_cmpx_ne_u32 vcc, %0, %1\n\tv_cmpx_ne_u32 vcc, %0, %1\n\t"
                              "v_cmpx_ne_u32 vcc, %0, %1\n\tv_cmpx_ne_u32 vcc, %0, %1\n\tv_cmpx_ne_u32 vcc, %0, %1\n\tv_cmpx_ne_u32 vcc, %0, %1\n\t")
                         "s_mov_b64 exec, s[10:11]\n\t"
                         : "+v"(a)
                         : "v"(0xFFFFFFFFu)
                         : "vcc", "s10", "s11");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678u) out[1] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, uint64_t* d) {
    const int iters = 2000;
    for (int threads : {256, 512, 1024}) {
        for (int wgs_per_cu : {1, 2}) {
            if (threads * wgs_per_cu > 2048) continue;
            hipLaunchKernelGGL(k<KIND>, dim3(256 * wgs_per_cu), dim3(threads), 0, 0, d, 10, 1u);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<KIND>, dim3(256 * wgs_per_cu), dim3(threads), 0, 0, d, iters, 1u);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            uint64_t cyc;
            hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
            const int waves_per_simd = threads / 64 / 4 * wgs_per_cu;
            const double instr_per_wave = (double)iters * 64.0;
            // s_memtime ticks at a constant 100 MHz: cycles from the event time at the boost clock is no better -- report both
            printf("%-22s %d waves/SIMD: %.3f ms for %.0f instr per wave -> %.2f ns per instr per SIMD (%.2f cycles at 2.4 GHz), memtime ticks %llu\n",
                   name, waves_per_simd, ms, instr_per_wave, ms * 1e6 / (instr_per_wave * waves_per_simd),
                   ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4, (unsigned long long)cyc);
        }
    }
}

int main() {
    uint64_t* d;
    hipMalloc(&d, 64);
    run<0>("8 independent v_add", d);
    run<1>("dependent v_add", d);
    run<2>("service mix", d);
    run<3>("v_cmpx chain", d);
    return 0;
}
