// pt_math.h — numerics contract of the MI355X radiance() path, shared by host and device code.
//
// Every function here is a strict per-operation IEEE-754 binary32 (or binary64 inside sin/cos)
// evaluation in the order the reference evaluates it; the translation units that include this
// header are built with -ffp-contract=off -fno-fast-math so neither hipcc nor the host compiler
// fuses or reassociates anything (rustc/LLVM never does for the reference either).
//
// Reference semantics restated (citations relative to /root/reference):
//   glam 0.30.8 scalar Vec3 (Cargo.lock:1508)  dot, cross, length, normalize = v*(1/len)
//   rand 0.8.5 Standard<f32> (mod.rs:53)       (u32 >> 8) * 2^-24
//   f32::sin / f32::cos (mod.rs:703)            libm sinf/cosf; glibc's double-polynomial algorithm,
//                                              bit-identical to glibc 2.35 on all 2^24 reachable
//                                              arguments (tests/test_oracle.py)
//   powi(2), powi(5) (mod.rs:416,741,754)      x*x, x*((x*x)*(x*x))
// rand01() itself (ThreadRng, OS seeded) is replaced by Philox4x32-7 keyed per
// (seed, pixel, sample, branch, depth) so that CPU and GPU draw the same numbers in any order.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define PT_HD static inline
#endif

namespace pt {

struct vec3 {
    float x, y, z;
};

PT_HD vec3 mk(float x, float y, float z) {
    vec3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
PT_HD vec3 operator+(vec3 a, vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD vec3 operator-(vec3 a, vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD vec3 operator*(vec3 a, vec3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD vec3 operator*(vec3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PT_HD vec3 operator/(vec3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
PT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_HD vec3 cross(vec3 a, vec3 b) {
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
PT_HD float f_abs(float x) { return __builtin_fabsf(x); }
// IEEE correctly rounded square root and reciprocal on both sides (x86: sqrtss / divss).
// On the device they are SHORT sequences around the native approximations, each proven equal to the compiler's IEEE
// expansion (-fhip-fp32-correctly-rounded-divide-sqrt) on every input of its fast-path domain by exhaustive comparison
// on the hardware (tools/rounding_search.hip over all 2^32 bit patterns -> profiles/r03_rounding_search.json; the same
// sweep runs through the library in the GPU tests: pt_ctx_numerics_sweep):
//   f_sqrt: v_rsq_f32, s = x*r, h = r/2, one residual d = fma(-s, s, x), result fma(d, h, s) - 16 issue cycles instead of the
//           32 of v_sqrt_f32 + two-neighbour residual test (round 1-2) - exact for every x in [2^-96, largest float]; negative
//           x gives NaN as IEEE wants.  Zero, denormals, normals below 2^-96 (the residual would underflow: a tangent ray's
//           discriminant can be that small), infinities and NaN take the compiler's full path: the whole wave does when
//           some lane holds one (one integer range test on |x|'s bits per call).
//   f_rcp : v_rcp_f32 and ONE Newton step - exact for every normal d with 2^-126 <= |d| <= 2^126 (round 1-2 ran LLVM's
//           division with numerator 1, seven operations).  Callers guarantee a normal in-range divisor wherever the result
//           is used (vector lengths of scene-scale vectors; determinants that passed |det| >= 1e-4).
// HIP's __fsqrt_rn / __frcp_rn are the 1-ulp native ops and are not usable (tests caught it).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float f_sqrt(float x) {
    // fast path iff 2^-96 <= |x| < inf: (bits(|x|) - bits(2^-96)) < (bits(inf) - bits(2^-96)) as unsigned integers
    const uint32_t off = (__float_as_uint(x) & 0x7fffffffu) - 0x0f800000u;
    if (__builtin_amdgcn_ballot_w64(off >= 0x70000000u) != 0ull) return __builtin_sqrtf(x);
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = x * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float f_rcp(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
#else
PT_HD float f_sqrt(float x) { return __builtin_sqrtf(x); }
PT_HD float f_rcp(float d) { return 1.0f / d; }
#endif
PT_HD float f_max(float a, float b) { return __builtin_fmaxf(a, b); }
PT_HD float length(vec3 a) { return f_sqrt(dot(a, a)); }
PT_HD vec3 normalize(vec3 a) { return a * f_rcp(length(a)); }

// ---- rand 0.8.5: 24 high bits of a u32 -> [0,1)
PT_HD float unit_f32(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

// ---- Philox4x32-R (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11); Random123's known-answer vectors
// for 7 and 10 rounds in tests/test_oracle.py.  The path draws with SEVEN rounds (Random123's philox4x32_R(7, ..): the
// fewest rounds the paper reports as Crush-resistant - it passes BigCrush - and the count its authors recommend where speed
// matters; ten is their default with a safety margin).  rand01() of the reference is an OS-seeded ThreadRng (mod.rs:47-55):
// which generator stands in for it is this build's own contract, and the generator was 5 % of k_pass_cand's time with ten
// rounds (profiles/r03_k_pass_cand_phase_budget.json; seven: +1.9 % frame rate, A/B on one GPU).
struct u32x4 {
    uint32_t a, b, c, d;
};
constexpr int kPhiloxRounds = 7;

template <int ROUNDS>
PT_HD u32x4 philox4x32(u32x4 ctr, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        // one 32x32->64 multiply per word pair (v_mad_u64_u32 on gfx950) instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.a, p1 = (uint64_t)0xCD9E8D57u * ctr.c;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        u32x4 nx;
        nx.a = hi1 ^ ctr.b ^ k0;
        nx.b = lo1;
        nx.c = hi0 ^ ctr.d ^ k1;
        nx.d = lo0;
        ctr = nx;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ctr;
}

// RNG contract: counter = (pixel index in the frame, sample index, tag, 0), key = seed.
//   tag 0                      : camera draws of a sample (a -> r1, b -> r2; mod.rs:818-819)
//   tag (branch<<8)|new_depth  : one radiance() invocation (a -> roulette mod.rs:678,
//                                b -> diffuse r1 / refract choice mod.rs:691,761, c -> diffuse r2 mod.rs:692)
PT_HD u32x4 draw_block(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t tag) {
    u32x4 c;
    c.a = pixel;
    c.b = sample;
    c.c = tag;
    c.d = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    // The key is the same for every draw of a kernel, so the compiler hoists the whole key schedule (k + r * W, fourteen
    // scalars) out of the kernels' loops - and, the pass kernels being short of SGPRs, parks it in VGPR lanes: fourteen
    // v_readlane per block, VALU issue slots each.  An opaque pass through the scalar registers keeps the schedule where it
    // is used: fourteen s_add per block, which issue beside the vector instructions.
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
    return philox4x32<kPhiloxRounds>(c, k0, k1);
}

// ---- sinf / cosf for |y| < 120: glibc (ARM optimized-routines) algorithm, evaluated in binary64
namespace sc {
constexpr double kHpiInv = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
constexpr double kHpi = 0x1.921FB54442D18p0;
constexpr double kC0 = 0x1p0, kC1 = -0x1.ffffffd0c621cp-2, kC2 = 0x1.55553e1068f19p-5,
                 kC3 = -0x1.6c087e89a359dp-10, kC4 = 0x1.99343027bf8c3p-16;
constexpr double kS1 = -0x1.555545995a603p-3, kS2 = 0x1.1107605230bc4p-7, kS3 = -0x1.994eb3774cf24p-13;

PT_HD uint32_t top12(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return (u >> 20) & 0x7ffu;
}
PT_HD float sin_poly(double x, double x2) {
    const double x3 = x * x2;
    const double s1 = kS2 + x2 * kS3;
    const double x7 = x3 * x2;
    const double s = x + x3 * kS1;
    return (float)(s + x7 * s1);
}
// `flip` selects the negated cosine table (quadrants 2 and 3)
PT_HD float cos_poly(double x2, bool flip) {
    const double g = flip ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = g * kC3 + x2 * (g * kC4);
    const double c1 = g * kC1 + x2 * (g * kC2);
    const double x6 = x4 * x2;
    const double c = g * kC0 + x2 * c1;
    return (float)(c + x6 * c2);
}
}  // namespace sc

// Sine and cosine of one argument, sharing the range reduction; valid for 0 <= y < 120.
// glibc keeps two shortcuts for small |y| (|y| < pi/4: no reduction; |y| < 2^-12: return y / 1).  Both are
// arithmetic no-ops on this domain: for y < pi/4 the reduction yields n = 0 and x - 0*hpi = x exactly, and for
// y < 2^-12 the polynomials round back to y and 1.0f.  So one branch-free path gives bit-identical results
// (checked against the two-shortcut restatement in the oracle, and through it against the platform libm, on all
// 2^24 reachable arguments: tests/test_abi.py) and the wave never runs two copies of the polynomials.
PT_HD void sincos_f32(float y, float *s_out, float *c_out) {
    double x = (double)y;
    const double r = x * sc::kHpiInv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * sc::kHpi;
    const int q = n & 3;
    const double sgn = (q == 1 || q == 2) ? -1.0 : 1.0;
    const float sp = sc::sin_poly(x * sgn, x * x);
    const float cp = sc::cos_poly(x * x, (n & 2) != 0);
    // sinf uses polynomial index n, cosf uses n^1: even -> sine polynomial, odd -> cosine polynomial
    const bool even = (n & 1) == 0;
    *s_out = even ? sp : cp;
    *c_out = even ? cp : sp;
}

// tent filter of render_pixel (mod.rs:820-830)
// (one square root, of the operand the branch taken would use: the same operations on the same values)
PT_HD float tent(float r) {
    const bool lo = r < 1.0f;
    const float s = f_sqrt(lo ? r : 2.0f - r);
    return lo ? s - 1.0f : 1.0f - s;
}

PT_HD float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

}  // namespace pt
