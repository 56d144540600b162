// dorado_amd/csrc/detmath.h — deterministic exp/log for the CRF decoder.
//
// The decoder's integer outputs (moves, bases) depend on float comparisons of log-sum-exp
// values, so to be bit-comparable with a CPU checker the transcendental kernel must be a fixed
// sequence of IEEE operations rather than whatever libm/ocml happens to do.  These are plain
// Cephes-style polynomials evaluated ONLY with fmaf/mul/add (compile this TU with
// -ffp-contract=off); the CPU oracle restates the same operation sequence independently.
// Accuracy ~1 ulp-class (max rel err < 4e-7), the same class as the glibc/Sleef routines the
// reference's CPU path uses (decode/CPUDecoder.cpp:34, decode/beam_search.cpp:42-45).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define DM_FN __device__ __forceinline__
#else
#define DM_FN static inline
#endif

DM_FN float dm_bits_to_f(uint32_t u) { return __builtin_bit_cast(float, u); }
DM_FN uint32_t dm_f_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// exp(x), intended for x <= 0; 0 below -103, clamps above 88.
DM_FN float dm_expf(float x) {
    if (x < -103.0f) {
        return 0.0f;
    }
    if (x > 88.0f) {
        x = 88.0f;
    }
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = __builtin_fmaf(p, r2, r);
    p = p + 1.0f;
    const int ni = (int)n;
    const int n1 = ni / 2, n2 = ni - n1;
    p = p * dm_bits_to_f((uint32_t)(n1 + 127) << 23);
    p = p * dm_bits_to_f((uint32_t)(n2 + 127) << 23);
    return p;
}

// log(x) for finite normal x > 0.
DM_FN float dm_logf(float x) {
    uint32_t ix = dm_f_to_bits(x);
    int e = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u;
    float m = dm_bits_to_f(ix);
    if (m > 1.41421356237f) {
        m = m * 0.5f;
        e += 1;
    }
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
    const float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// log-sum-exp of {stay, 4 steps}: max, sum of exp in argument order, log
// (at::logsumexp semantics, decode/CPUDecoder.cpp:28-34).
DM_FN float dm_lse5(float v0, float v1, float v2, float v3, float v4) {
    float m = v0;
    m = v1 > m ? v1 : m;
    m = v2 > m ? v2 : m;
    m = v3 > m ? v3 : m;
    m = v4 > m ? v4 : m;
    float s = dm_expf(v0 - m);
    s += dm_expf(v1 - m);
    s += dm_expf(v2 - m);
    s += dm_expf(v3 - m);
    s += dm_expf(v4 - m);
    return m + dm_logf(s);
}

// decode/beam_search.cpp:42-45
DM_FN float dm_log_sum_exp2(float x, float y) {
    const float d = __builtin_fabsf(x - y);
    const float m = x > y ? x : y;
    if (!(d < 17.0f)) {
        return m + 0.0f;
    }
    return m + dm_logf(1.0f + dm_expf(-d));
}
