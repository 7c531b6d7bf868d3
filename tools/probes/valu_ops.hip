// Development probe (GPU box): cycles a wave64 instruction holds its SIMD, opcode by opcode (gfx950, four waves per SIMD, eight
// independent chains per wave).  tools/probes/valu_rate.hip showed v_add_u32 at ~2.5 cycles and a mix of the service's opcodes at
// ~4.2: which ones are the slow ones decides how the hot blocks of k_match3 should be written.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_ops tools/probes/valu_ops.hip && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
// one instruction pattern per kernel: I(r) is the instruction on chain register %r; %8 = a vector operand, %9 = a scalar one
#define CHAINS(I) I("0") I("1") I("2") I("3") I("4") I("5") I("6") I("7")
#define DEF(NAME, I)                                                                                                   \
    __global__ void NAME(uint64_t* out, int iters, uint32_t seed) {                                                    \
        uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 77u, d = a ^ 0x55u, e = a + 5u, f = a ^ 9u, g = a + 11u, \
                 h = a ^ 13u;                                                                                          \
        const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)seed);                                       \
        asm volatile("s_mov_b64 vcc, 0x5555\n\ts_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");                     \
        for (int i = 0; i < iters; i++)                                                                                \
            asm volatile(REP8(CHAINS(I))                                                                               \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)                        \
                         : "v"(seed), "s"(sv)                                                                          \
                         : "vcc", "s20", "s21");                                                                         \
        if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678u) out[1] = a;                                                \
    }

#define I_ADD(r) "v_add_u32 %" r ", %" r ", %8\n\t"
#define I_SUB(r) "v_sub_u32 %" r ", %" r ", %8\n\t"
#define I_XOR(r) "v_xor_b32 %" r ", %" r ", %8\n\t"
#define I_OR(r) "v_or_b32 %" r ", %" r ", %8\n\t"
#define I_AND(r) "v_and_b32 %" r ", %" r ", %8\n\t"
#define I_MOV(r) "v_mov_b32 %" r ", %8\n\t"
#define I_LSHR(r) "v_lshrrev_b32 %" r ", 3, %" r "\n\t"
#define I_LSHL(r) "v_lshlrev_b32 %" r ", 1, %" r "\n\t"
#define I_MIN(r) "v_min_u32 %" r ", %" r ", %8\n\t"
#define I_CNDV(r) "v_cndmask_b32 %" r ", %" r ", %8, vcc\n\t"
#define I_CNDS(r) "v_cndmask_b32_e64 %" r ", %" r ", %8, s[20:21]\n\t"
#define I_PERM(r) "v_perm_b32 %" r ", %" r ", %8, %9\n\t"
#define I_ALIGN(r) "v_alignbyte_b32 %" r ", %" r ", %8, %9\n\t"
#define I_FFBL(r) "v_ffbl_b32 %" r ", %" r "\n\t"
#define I_CMP(r) "v_cmp_eq_u32 vcc, %" r ", %8\n\t"
#define I_CMPS(r) "v_cmp_eq_u32_e64 s[20:21], %" r ", %8\n\t"
#define I_SDWA(r) "v_add_u32_sdwa %" r ", %" r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define I_DPP(r) "v_mov_b32_dpp %" r ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_BFE(r) "v_bfe_u32 %" r ", %" r ", 3, 5\n\t"
#define I_ANDOR(r) "v_and_or_b32 %" r ", %" r ", %8, %9\n\t"
#define I_ADD3(r) "v_add3_u32 %" r ", %" r ", %8, %9\n\t"
#define I_LSHLADD(r) "v_lshl_add_u32 %" r ", %" r ", 1, %8\n\t"
#define I_XAD(r) "v_xad_u32 %" r ", %" r ", -1, %8\n\t"
#define I_MIN3(r) "v_min3_u32 %" r ", %" r ", %8, %9\n\t"
#define I_ADDLSHL(r) "v_add_lshl_u32 %" r ", %" r ", %8, 1\n\t"
#define I_MUL(r) "v_mul_lo_u32 %" r ", %" r ", %8\n\t"
#define I_MUL24(r) "v_mul_u32_u24 %" r ", %" r ", %8\n\t"
#define I_ADDE64(r) "v_add_u32_e64 %" r ", %" r ", %9\n\t"
#define I_PKADD(r) "v_pk_add_u16 %" r ", %" r ", %8\n\t"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 %" r ", %9, %" r "\n\t"
#define I_LSHLOR(r) "v_lshl_or_b32 %" r ", %" r ", 8, %8\n\t"
#define I_CNDV64(r) "v_cndmask_b32_e64 %" r ", %" r ", %8, vcc\n\t"
#define I_CMPCND(r) "v_cmp_eq_u32 vcc, %" r ", %8\n\tv_cndmask_b32 %" r ", %" r ", %8, vcc\n\t"
#define I_CMPSCNDS(r) "v_cmp_eq_u32_e64 s[20:21], %" r ", %8\n\tv_cndmask_b32_e64 %" r ", %" r ", %8, s[20:21]\n\t"
#define I_CMPX(r) "v_cmpx_ne_u32 vcc, %" r ", %8\n\t"
#define I_RFL(r) "v_readfirstlane_b32 s20, %" r "\n\t"
#define I_LSHLREV16(r) "v_lshlrev_b32 %" r ", 16, %" r "\n\t"
#define I_ADDLIT(r) "v_add_u32 %" r ", 0x12345, %" r "\n\t"
#define I_ANDLIT(r) "v_and_b32 %" r ", 0xffff, %" r "\n\t"
#define I_MAX(r) "v_max_u32 %" r ", %" r ", %8\n\t"
#define I_ADDCO(r) "v_add_co_u32 %" r ", vcc, %" r ", %8\n\t"
#define I_SUBREV(r) "v_subrev_u32 %" r ", %8, %" r "\n\t"
#define I_ASHR(r) "v_ashrrev_i32 %" r ", 3, %" r "\n\t"
#define I_NOT(r) "v_not_b32 %" r ", %" r "\n\t"
#define I_BFREV(r) "v_bfrev_b32 %" r ", %" r "\n\t"
#define I_FFBH(r) "v_ffbh_u32 %" r ", %" r "\n\t"
#define I_BCNT(r) "v_bcnt_u32_b32 %" r ", %" r ", %8\n\t"
#define I_XNOR(r) "v_xnor_b32 %" r ", %" r ", %8\n\t"
#define I_CVT(r) "v_cvt_f32_u32 %" r ", %" r "\n\t"
#define I_FADD(r) "v_add_f32 %" r ", %" r ", %8\n\t"
#define I_FMA(r) "v_fmac_f32 %" r ", %" r ", %8\n\t"
#define I_MOVS(r) "v_mov_b32 %" r ", %9\n\t"

DEF(k_add, I_ADD) DEF(k_sub, I_SUB) DEF(k_xor, I_XOR) DEF(k_or, I_OR) DEF(k_and, I_AND) DEF(k_mov, I_MOV) DEF(k_lshr, I_LSHR)
DEF(k_lshl, I_LSHL) DEF(k_min, I_MIN) DEF(k_cndv, I_CNDV) DEF(k_cnds, I_CNDS) DEF(k_perm, I_PERM) DEF(k_align, I_ALIGN)
DEF(k_ffbl, I_FFBL) DEF(k_cmp, I_CMP) DEF(k_cmps, I_CMPS) DEF(k_sdwa, I_SDWA) DEF(k_dpp, I_DPP) DEF(k_bfe, I_BFE)
DEF(k_andor, I_ANDOR) DEF(k_add3, I_ADD3) DEF(k_lshladd, I_LSHLADD) DEF(k_xad, I_XAD) DEF(k_min3, I_MIN3)
DEF(k_addlshl, I_ADDLSHL) DEF(k_mul, I_MUL) DEF(k_mul24, I_MUL24) DEF(k_adde64, I_ADDE64) DEF(k_pkadd, I_PKADD)
DEF(k_mbcnt, I_MBCNT) DEF(k_lshlor, I_LSHLOR)
DEF(k_cndv64, I_CNDV64) DEF(k_cmpcnd, I_CMPCND) DEF(k_cmpscnds, I_CMPSCNDS) DEF(k_rfl, I_RFL) DEF(k_lshl16, I_LSHLREV16)
DEF(k_addlit, I_ADDLIT) DEF(k_andlit, I_ANDLIT) DEF(k_max, I_MAX) DEF(k_addco, I_ADDCO) DEF(k_subrev, I_SUBREV) DEF(k_ashr, I_ASHR)
DEF(k_not, I_NOT) DEF(k_bfrev, I_BFREV) DEF(k_ffbh, I_FFBH) DEF(k_bcnt, I_BCNT) DEF(k_xnor, I_XNOR) DEF(k_cvt, I_CVT) DEF(k_fadd, I_FADD)
DEF(k_fma, I_FMA) DEF(k_movs, I_MOVS)

typedef void (*kern_t)(uint64_t*, int, uint32_t);
static double time_of(kern_t kf, uint64_t* d, int threads) {
    const int iters = 2000;
    hipLaunchKernelGGL(kf, dim3(256), dim3(threads), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kf, dim3(256), dim3(threads), 0, 0, d, iters, 1u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 64.0 * (threads / 256));  // ns per instruction per SIMD
}

int main() {
    uint64_t* d;
    hipMalloc(&d, 64);
    struct {
        const char* name;
        kern_t f;
    } ks[] = {{"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_xor_b32", k_xor}, {"v_or_b32", k_or}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov},
              {"v_lshrrev_b32", k_lshr}, {"v_lshlrev_b32", k_lshl}, {"v_min_u32", k_min}, {"v_cndmask vcc", k_cndv}, {"v_cndmask_e64 sgpr", k_cnds},
              {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_align}, {"v_ffbl_b32", k_ffbl}, {"v_cmp_eq vcc", k_cmp}, {"v_cmp_eq_e64 sgpr", k_cmps},
              {"v_add_u32_sdwa", k_sdwa}, {"v_mov_b32_dpp", k_dpp}, {"v_bfe_u32", k_bfe}, {"v_and_or_b32", k_andor}, {"v_add3_u32", k_add3},
              {"v_lshl_add_u32", k_lshladd}, {"v_xad_u32", k_xad}, {"v_min3_u32", k_min3}, {"v_add_lshl_u32", k_addlshl}, {"v_mul_lo_u32", k_mul},
              {"v_mul_u32_u24", k_mul24}, {"v_add_u32_e64 (sgpr src)", k_adde64}, {"v_pk_add_u16", k_pkadd}, {"v_mbcnt_lo", k_mbcnt},
              {"v_lshl_or_b32", k_lshlor}, {"v_cndmask_e64 vcc", k_cndv64}, {"v_cmp vcc + v_cndmask vcc (pair)", k_cmpcnd},
              {"v_cmp_e64 + v_cndmask_e64 (pair)", k_cmpscnds}, {"v_readfirstlane", k_rfl}, {"v_lshlrev_b32 16", k_lshl16},
              {"v_add_u32 literal", k_addlit}, {"v_and_b32 literal", k_andlit}, {"v_max_u32", k_max}, {"v_add_co_u32", k_addco},
              {"v_subrev_u32", k_subrev}, {"v_ashrrev_i32", k_ashr}, {"v_not_b32", k_not}, {"v_bfrev_b32", k_bfrev}, {"v_ffbh_u32", k_ffbh},
              {"v_bcnt_u32_b32", k_bcnt}, {"v_xnor_b32", k_xnor}, {"v_cvt_f32_u32", k_cvt}, {"v_add_f32", k_fadd}, {"v_fmac_f32", k_fma},
              {"v_mov_b32 sgpr", k_movs}};
    const double base = time_of(k_add, d, 1024);
    printf("%-28s %8s %8s %8s   (ns per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD; relative to v_add_u32 at 4)\n", "opcode", "1", "2", "4");
    for (auto& q : ks) {
        const double t1 = time_of(q.f, d, 256), t2 = time_of(q.f, d, 512), t4 = time_of(q.f, d, 1024);
        printf("%-28s %8.2f %8.2f %8.2f   x%.2f\n", q.name, t1, t2, t4, t4 / base);
    }
    return 0;
}
