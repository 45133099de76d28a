// valu_issue_bench.hip — cycles per wave-instruction per SIMD of the VALU opcodes the radiance() kernels issue,
// measured on the box (gfx950), at 1..8 waves per SIMD with every CU busy.
//
// Why: bench.py prices k_pass against a "VALU issue ceiling".  Round 1 assumed one wave-instruction per 4 cycles per
// SIMD for every opcode; /opt/skills/guides/MI355X_MICROARCH.md gives 2 cycles for a wave64 v_fma_f32 once more than one
// wave feeds the SIMD.  This program measures the table instead of assuming it.
//
// Method: one kernel per opcode.  Every wave runs ITERS trips of a block of 64 independent copies of the instruction
// (eight rotating destinations, sources never written: no RAW chain), stamps s_memtime (shader clock) and
// s_memrealtime (100 MHz) around the loop.  Launch shape pins the occupancy: W waves per SIMD = blocks of 256*Wb
// threads (Wb waves on each of the CU's 4 SIMDs), bpc blocks per CU enforced by the dynamic-LDS size, grid =
// 256 CUs * bpc: every wave of the grid is resident for the whole run.
//   cycles per wave-instruction per SIMD = median over waves of dt_memtime / (ITERS * 64 * W)
// (all W waves of a SIMD run concurrently for the same time, so the SIMD issued W * ITERS * 64 instructions in dt).
//
// build: hipcc -O2 --offload-arch=gfx950 -o valu_issue_bench tools/valu_issue_bench.hip
// run  : ./valu_issue_bench > gpurun_out/valu_issue_costs.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

extern __shared__ uint4 dyn_lds[];

// eight rotating destinations; D(n) = 32-bit dest vN, P(n) = 64-bit dest v[N:N+1]
#define R8(X) X(10, 11) X(12, 13) X(14, 15) X(16, 17) X(18, 19) X(20, 21) X(22, 23) X(24, 25)
#define R64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", \
             "v25", "v26", "v27", "v28", "v29", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", \
             "s25", "vcc", "scc", "memory"

struct Stamp {
    unsigned long long dt_clk, dt_real;
};

typedef float f2 __attribute__((ext_vector_type(2)));
// asm operands of the loop body (sources, never written): %0 %1 %2 = f32 VGPRs, %3 %4 = f64 VGPR pairs,
// %5 %6 = packed-f32 VGPR pairs, %7 = SGPR pair, %8 = SGPR
#define KERNEL(NAME, BODY)                                                                                       \
    __global__ void __launch_bounds__(1024) k_##NAME(Stamp *out, int iters, float fa, float fb, double da) {      \
        float s1 = fa + threadIdx.x * 1e-3f, s2 = fb + threadIdx.x * 1e-4f, s3 = fa * fb;                         \
        double d1 = da + threadIdx.x, d2 = da * 0.5;                                                               \
        f2 p1 = {s1, s2}, p2 = {s3, s1};                                                                           \
        unsigned long long sg64 = 0x400000003f800000ull;                                                           \
        unsigned sg32 = 0x3f800000u;                                                                               \
        asm volatile("" : "+v"(s1), "+v"(s2), "+v"(s3), "+v"(d1), "+v"(d2), "+v"(p1), "+v"(p2), "+s"(sg64), "+s"(sg32)); \
        unsigned long long t0, t1, r0, r1;                                                                         \
        __syncthreads();                                                                                           \
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");    \
        for (int i = 0; i < iters; ++i)                                                                            \
            asm volatile(BODY ::"v"(s1), "v"(s2), "v"(s3), "v"(d1), "v"(d2), "v"(p1), "v"(p2), "s"(sg64), "s"(sg32) : CLOB); \
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");    \
        if ((threadIdx.x & 63) == 0) {                                                                             \
            Stamp s;                                                                                               \
            s.dt_clk = t1 - t0;                                                                                    \
            s.dt_real = r1 - r0;                                                                                   \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                                                 \
        }                                                                                                          \
    }

// ---- the opcodes (v1,v2,v3,v8,v9: f32/u32; v[4:5], v[6:7]: f64; v[2:3], v[8:9] also serve as packed-f32 pairs)
#define OP_mul_f32(a, b) "v_mul_f32 v" #a ", %0, %1\n"
#define OP_add_f32(a, b) "v_add_f32 v" #a ", %0, %1\n"
#define OP_sub_f32_sgpr(a, b) "v_sub_f32 v" #a ", %8, %1\n"
#define OP_fma_f32(a, b) "v_fma_f32 v" #a ", %0, %1, %2\n"
#define OP_fma_f32_neg(a, b) "v_fma_f32 v" #a ", -%0, %1, %2\n"
#define OP_max_f32(a, b) "v_max_f32 v" #a ", %0, %1\n"
#define OP_mov_b32(a, b) "v_mov_b32 v" #a ", %0\n"
#define OP_pk_mul_f32(a, b) "v_pk_mul_f32 v[" #a ":" #b "], %5, %6\n"
#define OP_pk_add_f32(a, b) "v_pk_add_f32 v[" #a ":" #b "], %5, %6\n"
#define OP_pk_fma_f32(a, b) "v_pk_fma_f32 v[" #a ":" #b "], %5, %6, %5\n"
#define OP_pk_mul_f32_sgpr(a, b) "v_pk_mul_f32 v[" #a ":" #b "], %5, %7\n"
#define OP_pk_mov_b32(a, b) "v_pk_mov_b32 v[" #a ":" #b "], %5, %6\n"
#define OP_cndmask_b32(a, b) "v_cndmask_b32 v" #a ", %0, %1, vcc\n"
#define OP_cmp_lt_f32_vcc(a, b) "v_cmp_lt_f32 vcc, %0, %1\n"
#define OP_cmp_lt_f32_sgpr(a, b) "v_cmp_lt_f32 s[" #a ":" #b "], %0, %1\n"  /* s[10:11].. are free: kernel uses few SGPRs */
#define OP_cmp_class_f32(a, b) "v_cmp_class_f32 vcc, %0, %1\n"
#define OP_mad_u64_u32(a, b) "v_mad_u64_u32 v[" #a ":" #b "], s[20:21], %0, %1, %3\n"
#define OP_mul_lo_u32(a, b) "v_mul_lo_u32 v" #a ", %0, %1\n"
#define OP_mul_hi_u32(a, b) "v_mul_hi_u32 v" #a ", %0, %1\n"
#define OP_add_u32(a, b) "v_add_u32 v" #a ", %0, %1\n"
#define OP_xor_b32(a, b) "v_xor_b32 v" #a ", %0, %1\n"
#define OP_lshrrev_b32(a, b) "v_lshrrev_b32 v" #a ", 8, %0\n"
#define OP_lshl_or_b32(a, b) "v_lshl_or_b32 v" #a ", %0, 8, %1\n"
#define OP_and_or_b32(a, b) "v_and_or_b32 v" #a ", %0, %1, %2\n"
#define OP_bfe_u32(a, b) "v_bfe_u32 v" #a ", %0, 8, 10\n"
#define OP_cvt_f32_u32(a, b) "v_cvt_f32_u32 v" #a ", %0\n"
#define OP_cvt_u32_f32(a, b) "v_cvt_u32_f32 v" #a ", %0\n"
#define OP_cvt_f64_f32(a, b) "v_cvt_f64_f32 v[" #a ":" #b "], %0\n"
#define OP_cvt_f32_f64(a, b) "v_cvt_f32_f64 v" #a ", %3\n"
#define OP_cvt_i32_f64(a, b) "v_cvt_i32_f64 v" #a ", %3\n"
#define OP_cvt_f64_i32(a, b) "v_cvt_f64_i32 v[" #a ":" #b "], %0\n"
#define OP_fma_f64(a, b) "v_fma_f64 v[" #a ":" #b "], %3, %4, %3\n"
#define OP_mul_f64(a, b) "v_mul_f64 v[" #a ":" #b "], %3, %4\n"
#define OP_add_f64(a, b) "v_add_f64 v[" #a ":" #b "], %3, %4\n"
#define OP_rcp_f32(a, b) "v_rcp_f32 v" #a ", %0\n"
#define OP_sqrt_f32(a, b) "v_sqrt_f32 v" #a ", %0\n"
#define OP_rsq_f32(a, b) "v_rsq_f32 v" #a ", %0\n"
#define OP_readlane_b32(a, b) "v_readlane_b32 s" #a ", %0, 3\n"
#define OP_readfirstlane_b32(a, b) "v_readfirstlane_b32 s" #a ", %0\n"
#define OP_writelane_b32(a, b) "v_writelane_b32 v" #a ", %8, 3\n"
#define OP_mbcnt_lo(a, b) "v_mbcnt_lo_u32_b32 v" #a ", %8, %0\n"
#define OP_s_and_b64(a, b) "s_and_b64 s[" #a ":" #b "], %7, %7\n"
#define OP_s_mul_i32(a, b) "s_mul_i32 s" #a ", %8, %8\n"
#define OP_s_nop(a, b) "s_nop 0\n"
#define OP_sub_f32(a, b) "v_sub_f32 v" #a ", %0, %1\n"
#define OP_mul_f32_sgpr(a, b) "v_mul_f32 v" #a ", %8, %1\n"
#define OP_add_f32_sgpr(a, b) "v_add_f32 v" #a ", %8, %1\n"
#define OP_fma_f32_sgpr(a, b) "v_fma_f32 v" #a ", %8, %1, %2\n"
#define OP_mul_f32_inl(a, b) "v_mul_f32 v" #a ", 2.0, %1\n"
#define OP_mul_f32_lit(a, b) "v_mul_f32 v" #a ", 0x40490fdb, %1\n"
#define OP_mul_f32_abs(a, b) "v_mul_f32_e64 v" #a ", |%0|, %1\n"
#define OP_add_f32_neg(a, b) "v_add_f32_e64 v" #a ", -%0, %1\n"
#define OP_fmac_f32(a, b) "v_fmac_f32 v" #a ", %0, %1\n"
#define OP_min_f32(a, b) "v_min_f32 v" #a ", %0, %1\n"
#define OP_med3_f32(a, b) "v_med3_f32 v" #a ", %0, %1, %2\n"
#define OP_and_b32(a, b) "v_and_b32 v" #a ", %0, %1\n"
#define OP_or_b32(a, b) "v_or_b32 v" #a ", %0, %1\n"
#define OP_lshlrev_b32(a, b) "v_lshlrev_b32 v" #a ", 3, %0\n"
#define OP_sub_u32(a, b) "v_sub_u32 v" #a ", %0, %1\n"
#define OP_add_co_u32(a, b) "v_add_co_u32 v" #a ", vcc, %0, %1\n"
#define OP_mul_u32_u24(a, b) "v_mul_u32_u24 v" #a ", %0, %1\n"
#define OP_mad_u32_u24(a, b) "v_mad_u32_u24 v" #a ", %0, %1, %2\n"
#define OP_add3_u32(a, b) "v_add3_u32 v" #a ", %0, %1, %2\n"
#define OP_cndmask_e64(a, b) "v_cndmask_b32_e64 v" #a ", %0, %1, %7\n"
#define OP_cmp_gt_u32(a, b) "v_cmp_gt_u32 vcc, %0, %1\n"
#define OP_cmp_eq_u32_sgpr(a, b) "v_cmp_eq_u32 s[" #a ":" #b "], %0, %1\n"
#define OP_rndne_f32(a, b) "v_rndne_f32 v" #a ", %0\n"
#define OP_trunc_f32(a, b) "v_trunc_f32 v" #a ", %0\n"
#define OP_cvt_f32_i32(a, b) "v_cvt_f32_i32 v" #a ", %0\n"
#define OP_ldexp_f32(a, b) "v_ldexp_f32 v" #a ", %0, %1\n"
#define OP_bfi_b32(a, b) "v_bfi_b32 v" #a ", %0, %1, %2\n"
#define OP_perm_b32(a, b) "v_perm_b32 v" #a ", %0, %1, %2\n"
#define OP_mov_b32_sgpr(a, b) "v_mov_b32 v" #a ", %8\n"
#define OP_mov_b32_inl(a, b) "v_mov_b32 v" #a ", 1.0\n"
#define OP_pk_add_f32_neg(a, b) "v_pk_add_f32 v[" #a ":" #b "], %5, %6 neg_lo:[0,1] neg_hi:[0,1]\n"
#define OP_mix_mul_max(a, b) "v_mul_f32 v" #a ", %0, %1\n v_max_f32 v" #b ", %0, %1\n"
#define OP_mix_mul_cvt(a, b) "v_mul_f32 v" #a ", %0, %1\n v_cvt_f32_u32 v" #b ", %0\n"
#define OP_mix_mul_pkmul(a, b) "v_mul_f32 v" #a ", %0, %1\n v_pk_mul_f32 v[" #a ":" #b "], %5, %6\n"
#define OP_mix_fma_mad64(a, b) "v_fma_f32 v" #a ", %0, %1, %2\n v_mad_u64_u32 v[" #a ":" #b "], s[20:21], %0, %1, %3\n"
#define OP_mix_mul_f64(a, b) "v_mul_f32 v" #a ", %0, %1\n v_fma_f64 v[" #a ":" #b "], %3, %4, %3\n"
#define OP_mix_mul_rcp(a, b) "v_mul_f32 v" #a ", %0, %1\n v_rcp_f32 v" #b ", %0\n"
#define OP_mix_mul3_rcp(a, b) "v_mul_f32 v" #a ", %0, %1\n v_add_f32 v" #b ", %0, %1\n v_mul_f32 v" #a ", %0, %2\n v_rcp_f32 v" #b ", %0\n"
#define OP_mix_cmp_max(a, b) "v_cmp_lt_f32 vcc, %0, %1\n v_max_f32 v" #a ", %0, %1\n"
#define OP_mix_pk_cmp(a, b) "v_pk_mul_f32 v[" #a ":" #b "], %5, %6\n v_cmp_lt_f32 vcc, %0, %1\n"
#define OP_mix_mul_salu2(a, b) "v_mul_f32 v" #a ", %0, %1\n s_and_b64 s[20:21], %7, %7\n s_mul_i32 s22, %8, %8\n"
#define OP_mix_mul_readlane(a, b) "v_mul_f32 v" #a ", %0, %1\n v_readlane_b32 s20, %0, 3\n"
#define OP_mix_cmp_cnd_mul2(a, b) "v_cmp_lt_f32 vcc, %0, %1\n v_mul_f32 v" #b ", %0, %1\n v_cndmask_b32 v" #a ", %0, %1, vcc\n v_add_f32 v" #b ", %0, %2\n"
// pairing rules: which neighbours share a 4-cycle slot (B = 4-cycle class, A = 2-cycle class; dests of A: v26..v29)
#define OP_pair_BA_indep(a, b) "v_max_f32 v" #a ", %0, %1\n v_mul_f32 v26, %0, %1\n"
#define OP_pair_BA_raw(a, b) "v_max_f32 v" #a ", %0, %1\n v_mul_f32 v26, v" #a ", %1\n"
#define OP_pair_AB_raw(a, b) "v_mul_f32 v26, %0, %1\n v_max_f32 v" #a ", v26, %1\n"
#define OP_trip_BAA(a, b) "v_max_f32 v" #a ", %0, %1\n v_mul_f32 v26, %0, %1\n v_add_f32 v27, %0, %2\n"
#define OP_quad_BBAA(a, b) "v_max_f32 v" #a ", %0, %1\n v_min_f32 v" #b ", %0, %1\n v_mul_f32 v26, %0, %1\n v_add_f32 v27, %0, %2\n"
#define OP_quad_BAAA(a, b) "v_max_f32 v" #a ", %0, %1\n v_mul_f32 v26, %0, %1\n v_add_f32 v27, %0, %2\n v_sub_f32 v28, %0, %2\n"
#define OP_pair_pkA_indep(a, b) "v_pk_mul_f32 v[" #a ":" #b "], %5, %7\n v_mul_f32 v26, %0, %1\n"
#define OP_pair_pkA_raw(a, b) "v_pk_mul_f32 v[" #a ":" #b "], %5, %7\n v_mul_f32 v26, v" #a ", %1\n"
#define OP_quad_pkAApkAA_raw(a, b) "v_pk_mul_f32 v[" #a ":" #b "], %5, %7\n v_mul_f32 v26, v" #a ", %1\n v_mul_f32 v27, v" #b ", %1\n"
#define OP_pair_sgprA(a, b) "v_mul_f32 v" #a ", %8, %1\n v_add_f32 v26, %0, %1\n"
#define OP_pair_cndA(a, b) "v_cndmask_b32_e64 v" #a ", %0, %1, %7\n v_add_f32 v26, %0, %1\n"
#define OP_pair_cvtA(a, b) "v_cvt_f32_u32 v" #a ", %0\n v_add_f32 v26, %0, %1\n"
#define OP_pair_mad64A(a, b) "v_mad_u64_u32 v[" #a ":" #b "], s[20:21], %0, %1, %3\n v_xor_b32 v26, %0, %1\n"
#define OP_pair_f64A(a, b) "v_fma_f64 v[" #a ":" #b "], %3, %4, %3\n v_add_f32 v26, %0, %1\n"
#define OP_pair_BA_raw2(a, b) "v_max_f32 v" #a ", %0, %1\n v_mul_f32 v26, %0, %1\n v_min_f32 v" #b ", %0, %2\n v_mul_f32 v27, v" #a ", %1\n"
#define OP_pair_AA_fma(a, b) "v_fma_f32 v" #a ", %0, %1, %2\n v_mul_f32 v26, %0, %1\n"
#define OP_pair_cmpA(a, b) "v_cmp_lt_f32 vcc, %0, %1\n v_mul_f32 v26, %0, %1\n"
#define OP_pair_cmpsA(a, b) "v_cmp_lt_f32 s[" #a ":" #b "], %0, %1\n v_mul_f32 v26, %0, %1\n"
#define OP_pair_rcpA(a, b) "v_rcp_f32 v" #a ", %0\n v_mul_f32 v26, %0, %1\n"
#define OP_trip_rcpAA(a, b) "v_rcp_f32 v" #a ", %0\n v_mul_f32 v26, %0, %1\n v_add_f32 v27, %0, %1\n"
#define OP_pair_B_salu(a, b) "v_max_f32 v" #a ", %0, %1\n s_and_b64 s[20:21], %7, %7\n"
#define OP_trip_BA_salu(a, b) "v_max_f32 v" #a ", %0, %1\n s_and_b64 s[20:21], %7, %7\n v_mul_f32 v26, %0, %1\n"
#define OP_trip_B_salu_A(a, b) "v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, %7\n v_mul_f32 v26, %0, %1\n"
// mixes (pairs of instructions per slot: 128 instructions per block, reported per instruction)
#define OP_mix_fma_salu(a, b) "v_fma_f32 v" #a ", %0, %1, %2\n s_and_b64 s[20:21], %7, %7\n"
#define OP_mix_fma_pkfma(a, b) "v_fma_f32 v" #a ", %0, %1, %2\n v_pk_fma_f32 v[" #a ":" #b "], %5, %6, %5\n"
#define OP_mix_mul_cmp(a, b) "v_mul_f32 v" #a ", %0, %1\n v_cmp_lt_f32 vcc, %0, %1\n"
#define OP_mix_cmp_cndmask(a, b) "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 v" #a ", %0, %1, vcc\n"
// dependent chains (latency): each instruction consumes the previous result
#define OP_dep_fma_f32(a, b) "v_fma_f32 v10, v10, %1, %2\n"
#define OP_dep_pk_fma_f32(a, b) "v_pk_fma_f32 v[10:11], v[10:11], %6, %5\n"
#define OP_dep_mul_f32(a, b) "v_mul_f32 v10, v10, %1\n"

#define LIST(X)                                                                                                        \
    X(mul_f32, 1) X(add_f32, 1) X(sub_f32_sgpr, 1) X(fma_f32, 1) X(fma_f32_neg, 1) X(max_f32, 1) X(mov_b32, 1)          \
    X(pk_mul_f32, 1) X(pk_add_f32, 1) X(pk_fma_f32, 1) X(pk_mul_f32_sgpr, 1) X(pk_mov_b32, 1) X(cndmask_b32, 1)         \
    X(cmp_lt_f32_vcc, 1) X(cmp_lt_f32_sgpr, 1) X(cmp_class_f32, 1) X(mad_u64_u32, 1) X(mul_lo_u32, 1) X(mul_hi_u32, 1)  \
    X(add_u32, 1) X(xor_b32, 1) X(lshrrev_b32, 1) X(lshl_or_b32, 1) X(and_or_b32, 1) X(bfe_u32, 1)       \
    X(cvt_f32_u32, 1) X(cvt_u32_f32, 1) X(cvt_f64_f32, 1) X(cvt_f32_f64, 1) X(cvt_i32_f64, 1) X(cvt_f64_i32, 1)         \
    X(fma_f64, 1) X(mul_f64, 1) X(add_f64, 1) X(rcp_f32, 1) X(sqrt_f32, 1) X(rsq_f32, 1) X(readlane_b32, 1)             \
    X(readfirstlane_b32, 1) X(writelane_b32, 1) X(mbcnt_lo, 1) X(s_and_b64, 1) X(s_mul_i32, 1) X(s_nop, 1)              \
    X(sub_f32, 1) X(mul_f32_sgpr, 1) X(add_f32_sgpr, 1) X(fma_f32_sgpr, 1) X(mul_f32_inl, 1) X(mul_f32_lit, 1)           \
    X(mul_f32_abs, 1) X(add_f32_neg, 1) X(fmac_f32, 1) X(min_f32, 1) X(med3_f32, 1) X(and_b32, 1) X(or_b32, 1)             \
    X(lshlrev_b32, 1) X(sub_u32, 1) X(add_co_u32, 1) X(mul_u32_u24, 1) X(mad_u32_u24, 1) X(add3_u32, 1) X(cndmask_e64, 1)  \
    X(cmp_gt_u32, 1) X(cmp_eq_u32_sgpr, 1) X(rndne_f32, 1) X(trunc_f32, 1) X(cvt_f32_i32, 1) X(ldexp_f32, 1)               \
    X(bfi_b32, 1) X(perm_b32, 1) X(mov_b32_sgpr, 1) X(mov_b32_inl, 1) X(pk_add_f32_neg, 1)               \
    X(mix_mul_max, 2) X(mix_mul_cvt, 2) X(mix_mul_pkmul, 2) X(mix_fma_mad64, 2) X(mix_mul_f64, 2)       \
    X(mix_mul_rcp, 2) X(mix_mul3_rcp, 4) X(mix_cmp_max, 2) X(mix_pk_cmp, 2) X(mix_mul_salu2, 3) X(mix_mul_readlane, 2)     \
    X(mix_cmp_cnd_mul2, 4)                                                                                                 \
    X(pair_BA_indep, 2) X(pair_BA_raw, 2) X(pair_AB_raw, 2) X(trip_BAA, 3) X(quad_BBAA, 4) X(quad_BAAA, 4)                 \
    X(pair_pkA_indep, 2) X(pair_pkA_raw, 2) X(quad_pkAApkAA_raw, 3) X(pair_sgprA, 2) X(pair_cndA, 2) X(pair_cvtA, 2)       \
    X(pair_mad64A, 2) X(pair_f64A, 2) X(pair_BA_raw2, 4) X(pair_AA_fma, 2) X(pair_cmpA, 2) X(pair_cmpsA, 2)                \
    X(pair_rcpA, 2) X(trip_rcpAA, 3) X(pair_B_salu, 2) X(trip_BA_salu, 3) X(trip_B_salu_A, 3)                              \
    X(mix_fma_salu, 2) X(mix_fma_pkfma, 2) X(mix_mul_cmp, 2) X(mix_cmp_cndmask, 2) X(dep_fma_f32, 1)                    \
    X(dep_pk_fma_f32, 1) X(dep_mul_f32, 1)

#define DEFINE(NAME, PER) KERNEL(NAME, R64(OP_##NAME))
LIST(DEFINE)

typedef void (*kern_t)(Stamp *, int, float, float, double);
struct Entry {
    const char *name;
    kern_t fn;
    int per_slot;
};
#define ENTRY(NAME, PER) {#NAME, k_##NAME, PER},
static Entry g_list[] = {LIST(ENTRY)};

struct Shape {
    int waves_per_simd, wb, bpc;  // wb = waves per SIMD inside one block, bpc = blocks per CU
};

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const Shape shapes[] = {{1, 1, 1}, {2, 2, 1}, {3, 3, 1}, {4, 4, 1}, {6, 3, 2}, {8, 4, 2}};
    Stamp *d_out = nullptr;
    const size_t max_waves = (size_t)cus * 32;
    CHECK(hipMalloc((void **)&d_out, max_waves * sizeof(Stamp)));
    std::vector<Stamp> h(max_waves);
    printf("{\n  \"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"iters\": %d, \"block_instructions\": 64,\n",
           prop.name, prop.gcnArchName, cus, iters);
    printf("  \"unit\": \"shader cycles (s_memtime) per wave-instruction per SIMD; every CU busy; median over waves\",\n");
    printf("  \"costs\": {\n");
    const int n_entries = (int)(sizeof(g_list) / sizeof(g_list[0]));
    for (int e = 0; e < n_entries; ++e) {
        const Entry &en = g_list[e];
        printf("    \"%s\": {", en.name);
        for (size_t si = 0; si < sizeof(shapes) / sizeof(shapes[0]); ++si) {
            const Shape &sh = shapes[si];
            const int block = 256 * sh.wb;
            const size_t lds = (size_t)(160 * 1024) / sh.bpc - 1024;  // at most bpc blocks per CU
            CHECK(hipFuncSetAttribute((const void *)en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * sh.bpc;
            const size_t waves = (size_t)grid * block / 64;
            // warm-up launch (clocks, code fetch), then the measured one
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(block), lds, 0, d_out, iters / 4, 1.5f, 0.75f, 3.25);
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(block), lds, 0, d_out, iters, 1.5f, 0.75f, 3.25);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), d_out, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> cyc(waves), clk(waves);
            for (size_t w = 0; w < waves; ++w) {
                cyc[w] = (double)h[w].dt_clk / ((double)iters * 64.0 * en.per_slot * sh.waves_per_simd);
                clk[w] = (double)h[w].dt_clk / (double)h[w].dt_real * 0.1;  // GHz: memrealtime ticks at 100 MHz
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(clk.begin(), clk.end());
            // cross-check that does not depend on which waves are co-resident: 8 grids' worth of blocks in one launch,
            // timed with HIP events: ns per wave-instruction per SIMD over the whole chip
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            Stamp *d_big = nullptr;
            CHECK(hipMalloc((void **)&d_big, waves * 8 * sizeof(Stamp)));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(grid * 8), dim3(block), lds, 0, d_big, iters, 1.5f, 0.75f, 3.25);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipFree(d_big));
            CHECK(hipEventDestroy(e0));
            CHECK(hipEventDestroy(e1));
            const double agg_ns = (double)ms * 1e6 / ((double)waves * 8 * iters * 64.0 * en.per_slot / (cus * 4.0));
            printf("%s\"w%d\": {\"cyc\": %.3f, \"p10\": %.3f, \"p90\": %.3f, \"ghz\": %.3f, \"agg_ns\": %.4f, \"agg_cyc\": %.3f}",
                   si ? ", " : "", sh.waves_per_simd, cyc[waves / 2], cyc[waves / 10], cyc[waves * 9 / 10], clk[waves / 2],
                   agg_ns, agg_ns * clk[waves / 2]);
            fflush(stdout);
        }
        printf("}%s\n", e + 1 < n_entries ? "," : "");
    }
    printf("  }\n}\n");
    CHECK(hipFree(d_out));
    return 0;
}
