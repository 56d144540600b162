// dorado_amd/csrc/conv.hip — conv front-end of the LSTM-CRF models.
//
// Replaces the reference's torch Conv1d+activation stack (dorado/nn/ConvStack.cpp:146-163) and
// the Koi call sites host_convolution_f16 (ConvStack.cpp:220) for conv1/conv2.  conv3 runs as an
// implicit-im2col MFMA GEMM (gemm.hip), the same decomposition as ConvStack.cpp:241-261.
//
// conv12_kernel: conv1 (1 -> 16, w5, s1) and conv2 (16 -> 16, w5, s1) fused; conv1's output lives
// only in LDS (f32).  HBM-bound elementwise-ish work: reads 2 B/sample, writes 32 B/sample.
// Round 5: conv2's 1280 MACs per sample (94 % of the kernel's arithmetic; the kernel ran at 0.93 TB/s, fp32-VALU-bound) moved
// from v_fma_f32 onto the f32-INPUT matrix instruction v_mfma_f32_16x16x4_f32: same rate as the vector FMA, but on the matrix
// pipe, i.e. beside the VALU work of the other waves (conv1, 32 swish per sample, conversions), and its result is bit for
// bit a k-ordered fmaf chain starting from C (cdna_hip_programming.md 3, "FP32-input MFMA") — with k = tap * 16 + ci ascending
// that is exactly the accumulation order of the VALU loop it replaces: every output bit is unchanged.
// Output layout: a2p [N][Tpitch][16] f16 with `pad` zero rows in front of each chunk (and zero
// rows behind), so that conv3's im2col row for output step t is the contiguous span
// a2p[n][stride*t .. stride*t + W)[0..16) — no gather needed.
#include "common.h"

#define C12_TT 256   // output time steps per workgroup
#define C12_CH 16
#define C12_W 5
#define C12_ROW 17   // padded LDS row (floats): odd stride => conv1's per-row writes and the MFMA path's column reads are conflict-free
#define C12_ROWV 20  // ... of the VALU conv2 (debug build, A/B): 80 B stride => conflict-free ds_read_b128

// RAW: x holds raw int16 samples and `ss` the per-chunk (shift, scale) (ScalerNode fused);
// VAR: `smask` marks the samples that belong to a chunk (variable chunk sizes, 8f-3).  Both are compile-time
// so that the plain f16 / fixed-size instantiation keeps its scalar-register weight schedule.
typedef float float4c __attribute__((ext_vector_type(4)));

// MF: conv2 on v_mfma_f32_16x16x4_f32 (product); !MF: the round 1-4 VALU loop (debug build only, MIBC_CONV12_VALU=1).
template <int ACT1, int ACT2, bool RAW, bool VAR, bool MF = true>
__global__ __launch_bounds__(C12_TT) void conv12_kernel(
        const half_t *__restrict__ x,    // [N][T_in]
        const float *__restrict__ w1,    // [5][16]     (k, co)
        const float *__restrict__ b1,    // [16]
        const float *__restrict__ w2,    // [5][16][16] (k, ci, co)
        const float *__restrict__ b2,    // [16]
        half_t *__restrict__ a2p,        // [N][Tpitch][16]
        half_t *__restrict__ a1_tap,     // optional [N][T_in][16] (parity tap) or nullptr
        const float *__restrict__ ss,    // nullptr, or [N][2] (shift, scale): x is raw int16 and the
                                         // ScalerNode map f16((x - shift) / scale) is applied on the fly
        const uint32_t *__restrict__ smask,  // nullptr, or [N][(T_in + 31) / 32] sample bitmap (variable-chunk
                                             // mode): samples outside every chunk behave as zero PADDING at
                                             // each convolution level, exactly as if the chunk stood alone
        int T_in, int Tpitch, int pad) {
    __shared__ float xs[C12_TT + 8];
    __shared__ unsigned char xm[C12_TT + 8];   // VAR: sample t0 - 4 + i belongs to a chunk
    constexpr int ROW = MF ? C12_ROW : C12_ROWV;
    __shared__ __attribute__((aligned(16))) float o1[(C12_TT + 4) * ROW];
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * C12_TT;
    const int tid = threadIdx.x;
    const half_t *xn = x + (size_t)n * T_in;

    const int mwords = (T_in + 31) >> 5;
    auto in_mask = [&](int t) -> bool {
        if constexpr (!VAR) return true;
        return (smask[(size_t)n * mwords + (t >> 5)] >> (t & 31)) & 1u;
    };
    if constexpr (RAW) {
        // tensor_utils.cpp:89-142 semantics: f32 subtract, IEEE divide, round to f16 — the value the
        // reference's pipeline would have stored in the read before chunking
        const int16_t *xi = (const int16_t *)x + (size_t)n * T_in;
        const float shift = ss[2 * n], scale = ss[2 * n + 1];
        for (int i = tid; i < C12_TT + 8; i += C12_TT) {
            const int t = t0 - 4 + i;
            const bool ok = (t >= 0 && t < T_in && in_mask(t));
            xs[i] = ok ? (float)(half_t)(((float)xi[t] - shift) / scale) : 0.0f;
            if constexpr (VAR) xm[i] = ok;
        }
    } else {
        for (int i = tid; i < C12_TT + 8; i += C12_TT) {
            const int t = t0 - 4 + i;
            const bool ok = (t >= 0 && t < T_in && in_mask(t));
            xs[i] = ok ? (float)xn[t] : 0.0f;
            if constexpr (VAR) xm[i] = ok;
        }
    }
    __syncthreads();
    // conv1 at times t0-2 .. t0+TT+1 (row r <-> time t0-2+r); zero outside [0,T_in) = conv2's padding
    for (int r = tid; r < C12_TT + 4; r += C12_TT) {
        const int t = t0 - 2 + r;
        bool inside = (t >= 0 && t < T_in);
        if constexpr (VAR) inside = inside && xm[r + 2];   // time t <-> xs index t - (t0 - 4) = r + 2
        float acc[C12_CH];
#pragma unroll
        for (int c = 0; c < C12_CH; ++c) acc[c] = b1[c];
#pragma unroll
        for (int k = 0; k < C12_W; ++k) {
            const float xv = xs[r + k];  // time t + k - 2  -> xs index (t-2+k) - (t0-4) = r + k
#pragma unroll
            for (int c = 0; c < C12_CH; ++c) acc[c] = fmaf(w1[k * C12_CH + c], xv, acc[c]);
        }
#pragma unroll
        for (int c = 0; c < C12_CH; ++c) {
            const float v = inside ? act_apply(acc[c], ACT1) : 0.0f;
            o1[r * ROW + c] = v;
        }
        if (a1_tap != nullptr && inside && r >= 2 && r < C12_TT + 2) {
#pragma unroll
            for (int c = 0; c < C12_CH; ++c)
                a1_tap[((size_t)n * T_in + t) * C12_CH + c] = (half_t)act_apply(acc[c], ACT1);
        }
    }
    __syncthreads();
    if constexpr (MF) {
        // conv2 as out[co][t] = sum_k W[co][k] A1[k][t], k = tap * 16 + ci: per 16 positions x 16 channels 20 MFMAs of k = 4.
        // A (weights): lane holds W[co = lane & 15][k = 4 ks + (lane >> 4)] — 20 registers, loaded once.
        // B (conv1 out): lane holds o1[time t + tap - 2][ci] for position lane & 15, k as above -> one ds_read_b32 per MFMA.
        // D: lane holds channels 4 (lane >> 4) + r of position lane & 15 -> 8 bytes of the [t][16] f16 row.
        const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
        float wreg[20];
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) wreg[ks] = w2[(((ks >> 2) * C12_CH) + 4 * (ks & 3) + lq) * C12_CH + l15];
        const float4c bias4 = *(const float4c *)(b2 + 4 * lq);
#pragma unroll
        for (int tp = 0; tp < 4; tp += 2) {          // two tiles at a time: two independent accumulator chains
            const int p0 = (wave * 4 + tp) * 16 + l15, p1 = p0 + 16;     // positions inside the workgroup's 256
            float4c a0 = bias4, a1 = bias4;
#pragma unroll
            for (int ks = 0; ks < 20; ++ks) {
                const int tap = ks >> 2, ci = 4 * (ks & 3) + lq;
                const float b0 = o1[(p0 + tap) * ROW + ci], b1v = o1[(p1 + tap) * ROW + ci];
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ks], b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ks], b1v, a1, 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = h ? p1 : p0;
                const float4c a = h ? a1 : a0;
                const int t = t0 + p;
                uint32_t mk = 0xffffffffu;
                if constexpr (VAR) mk = xm[p + 4] ? 0xffffffffu : 0u;
                half4_t o;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[r] = (half_t)__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, act_apply(a[r], ACT2)) & mk);
                if (t < T_in) *(half4_t *)(a2p + ((size_t)n * Tpitch + pad + t) * C12_CH + 4 * lq) = o;
            }
        }
        return;
    }
    const int t = t0 + tid;
    if (t >= T_in) {
        return;
    }
    float acc[C12_CH];
#pragma unroll
    for (int c = 0; c < C12_CH; ++c) acc[c] = b2[c];
#pragma unroll
    for (int k = 0; k < C12_W; ++k) {
        // input time t + k - 2 -> row (t+k-2) - (t0-2) = tid + k
        const float4_t *row = (const float4_t *)&o1[(tid + k) * ROW];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4_t v = row[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = q * 4 + j;
                const float *w = w2 + (k * C12_CH + ci) * C12_CH;
#pragma unroll
                for (int c = 0; c < C12_CH; ++c) acc[c] = fmaf(w[c], v[j], acc[c]);
            }
        }
    }
    half8_t o0, o1v;
    // VAR: samples outside every chunk become +0 through a bit mask (a select here makes the compiler branch
    // around the whole conv2 accumulation and lose its scalar-register weight schedule)
    uint32_t mk = 0xffffffffu;
    if constexpr (VAR) mk = xm[tid + 4] ? 0xffffffffu : 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        o0[c] = (half_t)__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, act_apply(acc[c], ACT2)) & mk);
        o1v[c] = (half_t)__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, act_apply(acc[8 + c], ACT2)) & mk);
    }
    half8_t *dst = (half8_t *)(a2p + ((size_t)n * Tpitch + pad + t) * C12_CH);
    dst[0] = o0;
    dst[1] = o1v;
}

extern "C" int mibc_launch_conv12(hipStream_t s, const half_t *x, const float *w1, const float *b1,
                                  const float *w2, const float *b2, half_t *a2p, half_t *a1_tap, const float *ss,
                                  const uint32_t *smask, int N, int T_in, int Tpitch, int pad, int act1, int act2) {
    dim3 grid((T_in + C12_TT - 1) / C12_TT, N);
#ifdef MIBC_DEBUG_KERNELS
    static const int valu = MIBC_ENV_INT("MIBC_CONV12_VALU", 0);   // A/B: conv2 on the vector ALU (rounds 1-4)
#define LAUNCH4(A1, A2, R, V)                                                                          \
    do {                                                                                               \
        if (valu) hipLaunchKernelGGL((conv12_kernel<A1, A2, R, V, false>), grid, dim3(C12_TT), 0, s, x, w1, b1, w2, b2, a2p, a1_tap, ss, smask, T_in, Tpitch, pad); \
        else hipLaunchKernelGGL((conv12_kernel<A1, A2, R, V, true>), grid, dim3(C12_TT), 0, s, x, w1, b1, w2, b2, a2p, a1_tap, ss, smask, T_in, Tpitch, pad); \
    } while (0)
#else
#define LAUNCH4(A1, A2, R, V)                                                                          \
    hipLaunchKernelGGL((conv12_kernel<A1, A2, R, V>), grid, dim3(C12_TT), 0, s, x, w1, b1, w2, b2, a2p, a1_tap, ss, \
                       smask, T_in, Tpitch, pad)
#endif
#define LAUNCH(A1, A2)                                                                                 \
    do {                                                                                               \
        if (ss != nullptr && smask != nullptr) LAUNCH4(A1, A2, true, true);                            \
        else if (ss != nullptr) LAUNCH4(A1, A2, true, false);                                          \
        else if (smask != nullptr) LAUNCH4(A1, A2, false, true);                                       \
        else LAUNCH4(A1, A2, false, false);                                                            \
    } while (0)
    if (act1 == 0 && act2 == 0) {
        LAUNCH(0, 0);
    } else if (act1 == 1 && act2 == 1) {
        LAUNCH(1, 1);
    } else if (act1 == 0 && act2 == 1) {
        LAUNCH(0, 1);
    } else if (act1 == 1 && act2 == 0) {
        LAUNCH(1, 0);
    } else {
        return 1;
    }
#undef LAUNCH
#undef LAUNCH4
    return 0;
}
