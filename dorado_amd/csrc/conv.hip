// dorado_amd/csrc/conv.hip — conv front-end of the LSTM-CRF models.
//
// Replaces the reference's torch Conv1d+activation stack (dorado/nn/ConvStack.cpp:146-163) and
// the Koi call sites host_convolution_f16 (ConvStack.cpp:220) for conv1/conv2.  conv3 runs as an
// implicit-im2col MFMA GEMM (gemm.hip), the same decomposition as ConvStack.cpp:241-261.
//
// conv12_kernel: conv1 (1 -> 16, w5, s1) and conv2 (16 -> 16, w5, s1) fused; conv1's output lives
// only in LDS (f32).  HBM-bound elementwise-ish work: reads 2 B/sample, writes 32 B/sample.
// Output layout: a2p [N][Tpitch][16] f16 with `pad` zero rows in front of each chunk (and zero
// rows behind), so that conv3's im2col row for output step t is the contiguous span
// a2p[n][stride*t .. stride*t + W)[0..16) — no gather needed.
#include "common.h"

#define C12_TT 256   // output time steps per workgroup
#define C12_CH 16
#define C12_W 5
#define C12_ROW 20   // padded LDS row (floats): 80 B stride => conflict-free ds_read_b128

// RAW: x holds raw int16 samples and `ss` the per-chunk (shift, scale) (ScalerNode fused);
// VAR: `smask` marks the samples that belong to a chunk (variable chunk sizes, 8f-3).  Both are compile-time
// so that the plain f16 / fixed-size instantiation keeps its scalar-register weight schedule.
template <int ACT1, int ACT2, bool RAW, bool VAR>
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
    __shared__ __attribute__((aligned(16))) float o1[(C12_TT + 4) * C12_ROW];
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
            o1[r * C12_ROW + c] = v;
        }
        if (a1_tap != nullptr && inside && r >= 2 && r < C12_TT + 2) {
#pragma unroll
            for (int c = 0; c < C12_CH; ++c)
                a1_tap[((size_t)n * T_in + t) * C12_CH + c] = (half_t)act_apply(acc[c], ACT1);
        }
    }
    __syncthreads();
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
        const float4_t *row = (const float4_t *)&o1[(tid + k) * C12_ROW];
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
#define LAUNCH4(A1, A2, R, V)                                                                          \
    hipLaunchKernelGGL((conv12_kernel<A1, A2, R, V>), grid, dim3(C12_TT), 0, s, x, w1, b1, w2, b2, a2p, a1_tap, ss, \
                       smask, T_in, Tpitch, pad)
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
