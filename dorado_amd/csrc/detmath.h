// dorado_amd/csrc/detmath.h — deterministic exp/log for the CRF decoder.
//
// The decoder's integer outputs (moves, bases) depend on float comparisons of log-sum-exp
// values, so to be bit-comparable with a CPU checker the transcendental kernel must be a fixed
// sequence of IEEE operations rather than whatever libm/ocml happens to do.  These are plain
// Cephes-style polynomials evaluated ONLY with fmaf/mul/add (compile this TU with
// -ffp-contract=off); the CPU oracle restates the same operation sequence independently.
// Accuracy of a log-sum-exp built from them: ~2e-7 relative, the same class as the glibc/Sleef
// routines the reference's CPU path uses (decode/CPUDecoder.cpp:34, decode/beam_search.cpp:42-45).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define DM_FN __device__ __forceinline__
#else
#define DM_FN static inline
#endif

DM_FN float dm_bits_to_f(uint32_t u) { return __builtin_bit_cast(float, u); }
DM_FN uint32_t dm_f_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// exp(x) for x <= 0 (every use subtracts the running maximum first).  Arguments below -86 are
// clamped: the result (< 5e-38) is absorbed by any sum that also holds the exp(0) = 1 term.
// One-constant range reduction: the error it leaves, |n| * 2e-8 relative, only grows where the
// value itself (2^n) has stopped mattering to the sum.
DM_FN float dm_expf(float x) {
    x = __builtin_fmaxf(x, -86.0f);
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    const float r = __builtin_fmaf(n, -0.693147182464599609375f, x);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    // p in [0.70, 1.42], n in [-124, 0]: scaling by 2^n is an exponent-field add
    return dm_bits_to_f(dm_f_to_bits(p) + ((uint32_t)(int)n << 23));
}

// log(x) for finite normal x > 0.
DM_FN float dm_logf(float x) {
    // mantissa to [sqrt(1/2), sqrt(2)) by re-biasing the exponent field around sqrt(1/2)
    const uint32_t ix = dm_f_to_bits(x) + (0x3f800000u - 0x3f3504f3u);
    const int e = (int)(ix >> 23) - 127;
    const float m = dm_bits_to_f((ix & 0x007fffffu) + 0x3f3504f3u);
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, f, -1.1514610310e-1f);
    p = __builtin_fmaf(p, f, 1.1676998740e-1f);
    p = __builtin_fmaf(p, f, -1.2420140846e-1f);
    p = __builtin_fmaf(p, f, 1.4249322787e-1f);
    p = __builtin_fmaf(p, f, -1.6668057665e-1f);
    p = __builtin_fmaf(p, f, 2.0000714765e-1f);
    p = __builtin_fmaf(p, f, -2.4999993993e-1f);
    p = __builtin_fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    y = __builtin_fmaf(-0.5f, z, y);
    const float r = f + y;
    return __builtin_fmaf((float)e, 0.693147182464599609375f, r);
}

// log-sum-exp of {stay, 4 steps}: max, sum of exp in argument order, log
// (at::logsumexp semantics, decode/CPUDecoder.cpp:28-34).
DM_FN float dm_lse5(float v0, float v1, float v2, float v3, float v4) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(v0, v1), __builtin_fmaxf(v2, v3)), v4);
    float s = dm_expf(v0 - m);
    s += dm_expf(v1 - m);
    s += dm_expf(v2 - m);
    s += dm_expf(v3 - m);
    s += dm_expf(v4 - m);
    return m + dm_logf(s);
}

#ifdef __HIPCC__
// ---- two-lane forms for the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): element-wise the
//      SAME IEEE operation sequence as dm_expf / dm_logf / dm_lse5, so results are bit-identical to the scalar
//      forms (and to the oracle's restatement). ----
typedef float dm_f2 __attribute__((ext_vector_type(2)));
typedef int dm_i2 __attribute__((ext_vector_type(2)));
typedef unsigned int dm_u2 __attribute__((ext_vector_type(2)));

DM_FN dm_f2 dm2_fma(dm_f2 a, dm_f2 b, dm_f2 c) { return __builtin_elementwise_fma(a, b, c); }

DM_FN dm_f2 dm2_expf(dm_f2 x) {
    x = __builtin_elementwise_max(x, (dm_f2)(-86.0f));
    const dm_f2 n = __builtin_elementwise_rint(x * (dm_f2)(1.44269504088896341f));
    const dm_f2 r = dm2_fma(n, (dm_f2)(-0.693147182464599609375f), x);
    dm_f2 p = (dm_f2)(1.9875691500e-4f);
    p = dm2_fma(p, r, (dm_f2)(1.3981999507e-3f));
    p = dm2_fma(p, r, (dm_f2)(8.3334519073e-3f));
    p = dm2_fma(p, r, (dm_f2)(4.1665795894e-2f));
    p = dm2_fma(p, r, (dm_f2)(1.6666665459e-1f));
    p = dm2_fma(p, r, (dm_f2)(5.0000001201e-1f));
    p = dm2_fma(p, r, (dm_f2)(1.0f));
    p = dm2_fma(p, r, (dm_f2)(1.0f));
    const dm_i2 ni = __builtin_convertvector(n, dm_i2);
    const dm_u2 bits = __builtin_bit_cast(dm_u2, p) + (__builtin_bit_cast(dm_u2, ni) << 23);
    return __builtin_bit_cast(dm_f2, bits);
}

DM_FN dm_f2 dm2_logf(dm_f2 x) {
    const dm_u2 ix = __builtin_bit_cast(dm_u2, x) + (dm_u2)(0x3f800000u - 0x3f3504f3u);
    const dm_i2 e = __builtin_bit_cast(dm_i2, ix >> 23) - (dm_i2)(127);
    const dm_f2 m = __builtin_bit_cast(dm_f2, (ix & (dm_u2)(0x007fffffu)) + (dm_u2)(0x3f3504f3u));
    const dm_f2 f = m - (dm_f2)(1.0f);
    const dm_f2 z = f * f;
    dm_f2 p = (dm_f2)(7.0376836292e-2f);
    p = dm2_fma(p, f, (dm_f2)(-1.1514610310e-1f));
    p = dm2_fma(p, f, (dm_f2)(1.1676998740e-1f));
    p = dm2_fma(p, f, (dm_f2)(-1.2420140846e-1f));
    p = dm2_fma(p, f, (dm_f2)(1.4249322787e-1f));
    p = dm2_fma(p, f, (dm_f2)(-1.6668057665e-1f));
    p = dm2_fma(p, f, (dm_f2)(2.0000714765e-1f));
    p = dm2_fma(p, f, (dm_f2)(-2.4999993993e-1f));
    p = dm2_fma(p, f, (dm_f2)(3.3333331174e-1f));
    dm_f2 y = (f * z) * p;
    y = dm2_fma((dm_f2)(-0.5f), z, y);
    const dm_f2 r = f + y;
    return dm2_fma(__builtin_convertvector(e, dm_f2), (dm_f2)(0.693147182464599609375f), r);
}

DM_FN dm_f2 dm2_lse5(dm_f2 v0, dm_f2 v1, dm_f2 v2, dm_f2 v3, dm_f2 v4) {
    const dm_f2 m = __builtin_elementwise_max(
            __builtin_elementwise_max(__builtin_elementwise_max(v0, v1), __builtin_elementwise_max(v2, v3)), v4);
    dm_f2 s = dm2_expf(v0 - m);
    s += dm2_expf(v1 - m);
    s += dm2_expf(v2 - m);
    s += dm2_expf(v3 - m);
    s += dm2_expf(v4 - m);
    return m + dm2_logf(s);
}
#endif

// decode/beam_search.cpp:42-45
DM_FN float dm_log_sum_exp2(float x, float y) {
    const float d = __builtin_fabsf(x - y);
    const float m = x > y ? x : y;
    if (!(d < 17.0f)) {
        return m + 0.0f;
    }
    return m + dm_logf(1.0f + dm_expf(-d));
}
