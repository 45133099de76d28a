// rounding_search.hip — exhaustive check of shorter correctly-rounded sqrt / reciprocal sequences against the compiler's
// IEEE expansions (-fhip-fp32-correctly-rounded-divide-sqrt) over ALL 2^32 binary32 inputs, on the GPU.
//   hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o tools/rounding_search tools/rounding_search.hip
//   ./tools/rounding_search
// Prints, per candidate, the number of inputs whose result differs in bits from the IEEE result, split by input class.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float sq_llvm_nodenorm(float x) {  // LLVM's f32 sqrt lowering when denormals are flushed
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h = __builtin_fmaf(h, e, h);
    s = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float sq_short(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = x * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float sq_mid(float x) {  // one refinement of s only
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    const float h = 0.5f * r;
    const float d0 = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(d0, h, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float sq_native_fix(float x) {  // v_sqrt + one fma correction with h from rsq
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float rc4(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-d, r1, 1.0f);
    return __builtin_fmaf(e1, r1, r1);
}
__device__ __forceinline__ float rc4b(float d) {  // final correction with r0
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-d, r1, 1.0f);
    return __builtin_fmaf(e1, r0, r1);
}
__device__ __forceinline__ float rc2(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
__device__ __forceinline__ float rc6(float d) {  // the shipped f_rcp
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-d, r1, 1.0f);
    const float q1 = __builtin_fmaf(e1, r1, r1);
    const float e2 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e2, r1, q1);
}

// classes, sqrt: 0 = normal x >= 2^-96 (up to the largest float), 1 = normal x < 2^-96, 2 = denormal / zero, 3 = inf / NaN / negative
// classes, rcp:  0 = |x| in [2^-96, 2^96], 1 = other normal (|x| > 2^126 has a denormal reciprocal), 2 = denormal / zero, 3 = inf / NaN
__device__ __forceinline__ int cls_sqrt(float x) {
    const uint32_t b = __float_as_uint(x), a = b & 0x7fffffffu;
    if (a > 0x7f800000u || a == 0x7f800000u || (b >> 31 && a != 0u)) return 3;
    if (a < 0x00800000u) return 2;
    return x >= 0x1p-96f ? 0 : 1;  // 0: every normal x >= 2^-96 up to the largest float; 1: normal x < 2^-96
}
__device__ __forceinline__ int cls_rcp(float x) {
    const uint32_t a = __float_as_uint(x) & 0x7fffffffu;
    if (a >= 0x7f800000u) return 3;
    if (a < 0x00800000u) return 2;
    const float ax = __builtin_fabsf(x);
    return (ax >= 0x1p-96f && ax <= 0x1p96f) ? 0 : 1;  // 1: includes |d| > 2^126 where 1/d is denormal
}
__device__ __forceinline__ bool same(float a, float b) {
    const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    return x == y || ((x & 0x7fffffffu) > 0x7f800000u && (y & 0x7fffffffu) > 0x7f800000u);
}

__global__ void sweep(unsigned long long *bad) {  // bad[variant][class]
    unsigned long long local[8][4] = {};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float s_ref = __builtin_sqrtf(x), r_ref = 1.0f / x;
        const int cs = cls_sqrt(x), cr = cls_rcp(x);
        local[0][cs] += !same(sq_llvm_nodenorm(x), s_ref);
        local[1][cs] += !same(sq_short(x), s_ref);
        local[2][cs] += !same(sq_mid(x), s_ref);
        local[3][cs] += !same(sq_native_fix(x), s_ref);
        local[4][cr] += !same(rc4(x), r_ref);
        local[5][cr] += !same(rc4b(x), r_ref);
        local[6][cr] += !same(rc2(x), r_ref);
        local[7][cr] += !same(rc6(x), r_ref);
    }
    for (int v = 0; v < 8; ++v)
        for (int c = 0; c < 4; ++c)
            if (local[v][c]) atomicAdd(&bad[v * 4 + c], local[v][c]);
}

int main() {
    unsigned long long *d = nullptr, h[32];
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) return 1;
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(sweep, dim3(8192), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[8] = {"sqrt: rsq + LLVM flush-mode refinement (7 ops)", "sqrt: rsq + one fma correction (4 ops)",
                            "sqrt: rsq + two corrections of s (6 ops)", "sqrt: v_sqrt + rsq + one correction",
                            "rcp: 4 fma (final correction with r1)", "rcp: 4 fma (final correction with r0)", "rcp: 2 fma",
                            "rcp: 6 fma (shipped f_rcp)"};
    printf("{\n");
    for (int v = 0; v < 8; ++v)
        printf(" \"%s\": {\"mismatch_main_range\": %llu, \"mismatch_other_normal\": %llu, \"mismatch_denormal_zero\": %llu, \"mismatch_inf_nan_neg\": %llu}%s\n",
               names[v], h[v * 4], h[v * 4 + 1], h[v * 4 + 2], h[v * 4 + 3], v == 7 ? "" : ",");
    printf("}\n");
    return 0;
}
