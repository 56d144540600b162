// dorado_amd/csrc/lstm_q8.hip — the reference's QUANTISED LSTM path (opt-in: mibc_model_desc::lstm_quant).
//
// Reference (CUDA build): nn/LSTMStack.cpp:127-211 forward_cutlass with type_id KOI_I8 — the concatenated weights
// [W_ih | W_hh] are quantised per OUTPUT ROW with utils::quantize_tensor(weights, 1) (torch_utils/tensor_utils.cpp:293-300:
// scale = 128 / max|row|, round, clip +-127; LSTMStack.cpp:165-172), activations travel as int8 (working-memory layout
// CUTLASS_TNC_I8), the first layer runs in f16 when the convolutions hand over f16 and its output is converted to int8
// (host_convert, :199-207); bias and cell state stay floating point.  Koi itself is closed, so the activation scale is
// ours: h and the tanh output of conv3 lie in (-1, 1) and are stored as round(127 v).
//
// Kernel = lstm_layer_x8_kernel (lstm.hip) on v_mfma_i32_16x16x64_i8: the same 8 waves x 16-unit hidden tiles, x_t and
// h_{t-1} in LDS, weights streamed from L2 in fragment order through a register ring — but every operand byte count is
// halved (LDS fragment reads, L2 weight stream, HBM activations) and a k-step covers 64 instead of 32 inputs at the same
// MFMA cost: the three co-limiters of the f16 kernel (L2 port, LDS bytes, matrix pipe) all drop by 2.
//   gate pre-activation = float(acc_i32) * deq[row] + bias[row],   deq[row] = 1 / (127 * scale[row])
// Integer accumulation is exact; gates, cell state and the h quantisation are fp32 as in the f16 kernel.
#include "common.h"
#include "engine.h"

#ifndef Q8_STAGGER_DEFAULT
#define Q8_STAGGER_DEFAULT 0
#endif
#ifndef Q8_NT_DEFAULT
#define Q8_NT_DEFAULT 1   // measured: LSTM stack 195.1 -> 186.1 ms on the hac batch (profiles/r05_i_nt_ab_wsgemm_q8.log)
#endif
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4q __attribute__((ext_vector_type(4)));
typedef float float2q __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int4v q8_wload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ int q8_quant(float v) {   // round(127 v), |v| <= 1
    return (int)__builtin_rintf(v * 127.0f);
}

// OUT_F16: the last layer hands f16 to the CRF head (Xout f16 [T][N][C]); otherwise Xout is int8 [T][N][C].
// MASKED (variable chunk sizes, round 5 — the reference's default GPU mode is the quantised LSTM WITH variable chunks,
// basecall/CudaModelRunner.cpp:21-49 over nn/LSTMStack.cpp:127-211): bit r of tmask[t * gridDim.x + blockIdx.x] = "row r of this
// workgroup is inside a chunk at step t"; outside, h and c are forced to 0 exactly as in lstm_layer_x8_kernel<.., MASKED>.
// NTX: non-temporal x_t loads and h_t stores (as lstm_layer_x8_kernel, round 5); cache policy only.
template <int C, int PF, bool OUT_F16, bool MASKED = false, bool NTX = false>
__global__ __launch_bounds__(512, 2) void lstm_layer_q8_kernel(
        const int8_t *__restrict__ Xin,   // [T][N][C] int8 = round(127 x)
        void *__restrict__ Xout_,
        const int8_t *__restrict__ Wq,    // [C/16][2C/64][4][64][16]: lane (l15, lq): row g C + 16 j + l15, k = 64 ks + 16 lq ..
        const float *__restrict__ biasn,  // [4C]: [(hidden/32)][4][32]  (b_ih + b_hh)
        const float *__restrict__ deqn,   // [4C]: same order, 1 / (127 * row scale)
        int T, int N, int reverse, const unsigned long long *__restrict__ tmask = nullptr, int stagger = 0) {
    constexpr int NB = 64;
    constexpr int NT = 512;
    constexpr int HT = C / 16 / 8;    // 16-unit hidden tiles per wave
    constexpr int KS = 2 * C / 64;    // k-steps of 64 over [x ; h]
    constexpr int KSX = C / 64;
    constexpr int LDB = C + 32;       // bytes per LDS row: 16-byte slot stride = 2 mod 4 -> conflict-free ds_read_b128
    constexpr int LDH = C + 16;       // halfs per row of the f16 output staging (OUT_F16)
    constexpr int XPF = C / 128;      // 16-byte chunks per thread for one x_t block
    constexpr int KTOT = HT * KS;
    static_assert(KS % PF == 0 && (PF % 2) == 0, "k-steps must divide the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char q8_smem[];
    int8_t *hbuf0 = (int8_t *)q8_smem;
    int8_t *hbuf1 = hbuf0 + NB * LDB;
    int8_t *xbuf = hbuf1 + NB * LDB;
    float *bias_s = (float *)(xbuf + NB * LDB);
    float *deq_s = bias_s + 4 * C;
    half_t *hout = (half_t *)(deq_s + 4 * C);      // OUT_F16 only: [NB][LDH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * NB;

    for (int i = tid; i < NB * LDB / 16; i += NT) ((int4v *)hbuf0)[i] = (int4v)(0);
    // gate pre-activations are only ever used as exp2 arguments: sigmoid(p) = 1 / (1 + 2^(-log2e p)), tanh(p) = 2 / (1 + 2^(-2 log2e p)) - 1,
    // so bias and dequantisation factor are stored pre-multiplied by -log2e (gates i, f, o) / -2 log2e (gate g; order
    // [(hidden / 32)][gate][32]) and the fma that dequantises an accumulator yields the exp2 argument directly (round 6: the
    // kernel is bound by the vector-instruction count of this gate math, not by the matrix pipe — 33 + 10 quarter-rate
    // instructions per element against 12 MFMA cycles; profiles/r06_e_pmc_clock_hac_q8_n16384.json)
    for (int i = tid; i < 4 * C; i += NT) {
        const float k = (((i >> 5) & 3) == 2) ? -2.88539008f : -1.44269504f;
        bias_s[i] = biasn[i] * k;
        deq_s[i] = deqn[i] * k;
    }

    float4q cst[HT][4];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cst[a][b] = (float4q)(0.0f);

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(Wq + (size_t)wave * HT * KS * 4 * 64 * 16), 0, HT * KS * 4 * 64 * 16, 0x00020000);
    const int wvoff = lane * 16;
    int4v wr[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[u][g] = q8_wload(wrs, wvoff, (u * 4 + g) * 1024);
    int kpre = PF % KTOT;     // (C = 128: the ring holds the wave's whole k range)

    {
        const int t_first = reverse ? (T - 1) : 0;
        const int8_t *xg = Xin + ((size_t)t_first * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 16), col = c % (C / 16);
            *(int4v *)(xbuf + row * LDB + col * 16) = *(const int4v *)(xg + (size_t)c * 16);
        }
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? (T - 1 - step) : step;
        const int tn = (step + 1 < T) ? (reverse ? (t - 1) : (t + 1)) : t;
        const int8_t *hprev = (step & 1) ? hbuf1 : hbuf0;
        int8_t *hnext = (step & 1) ? hbuf0 : hbuf1;

        unsigned long long vm = ~0ull;
        if (MASKED) vm = tmask[(size_t)t * gridDim.x + blockIdx.x];
        // SIMD partners in anti-phase: waves w and w + 4 share a SIMD and, left alone, run their k-loops (matrix pipe) and their
        // gate math (vector pipe) at the same time, so the two pipes take turns.  Waves 4-7 start the step `stagger` x 64 cycles late
        // (about one k-loop): from then on one partner's gates run under the other's MFMAs.
        if (stagger > 0 && wave >= 4)
            for (int d = 0; d < stagger; ++d) __builtin_amdgcn_s_sleep(1);
        int4v xpf[XPF];
        {
            const int8_t *xg = Xin + ((size_t)tn * N + n0) * C;
#pragma unroll
            for (int p = 0; p < XPF; ++p)
                xpf[p] = NTX ? __builtin_nontemporal_load((const int4v *)(xg + (size_t)(tid + NT * p) * 16)) : *(const int4v *)(xg + (size_t)(tid + NT * p) * 16);
        }

#pragma unroll
        for (int jj = 0; jj < HT; ++jj) {
            const int j = wave * HT + jj;  // 16-unit hidden tile
            int4v acc[4][4];               // [gate][row tile of 16]
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[g][rt] = (int4v)(0);
            const int8_t *xb = xbuf + l15 * LDB + 16 * lq;
            const int8_t *hb = hprev + l15 * LDB + 16 * lq;
            int4v bq[2][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) bq[0][rt] = *(const int4v *)(xb + rt * 16 * LDB);
#pragma nounroll
            for (int ks0 = 0; ks0 < KS; ks0 += PF) {
#pragma unroll
                for (int uu = 0; uu < PF; ++uu) {
                    const int kn = (ks0 + uu + 1 < KS) ? (ks0 + uu + 1) : (KS - 1);
                    const int8_t *bn = (kn < KSX) ? (xb + kn * 64) : (hb + (kn - KSX) * 64);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) bq[(uu + 1) & 1][rt] = *(const int4v *)(bn + rt * 16 * LDB);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt)
                            acc[g][rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wr[uu][g], bq[uu & 1][rt], acc[g][rt], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) wr[uu][g] = q8_wload(wrs, wvoff, (kpre * 4 + g) * 1024);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                }
            }
            // gates: D row = hidden 16 j + 4 lq + r, D col = batch row l15 of row tile rt
            float4q bv[4], dv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = ((j >> 1) * 4 + g) * 32 + (j & 1) * 16 + 4 * lq;
                bv[g] = *(const float4q *)(bias_s + o);
                dv[g] = *(const float4q *)(deq_s + o);
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                half4_t hv;
                int pk = 0;
                // two hidden units at a time on the packed-f32 pipe (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two elements per
                // issue slot); only the conversions and the ten transcendentals per element stay scalar
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    float2q e[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float2q a = {(float)acc[g][rt][2 * h2], (float)acc[g][rt][2 * h2 + 1]};
                        const float2q dq = {dv[g][2 * h2], dv[g][2 * h2 + 1]};
                        const float2q bb = {bv[g][2 * h2], bv[g][2 * h2 + 1]};
                        const float2q x = __builtin_elementwise_fma(a, dq, bb);                 // the exp2 argument (pre-scaled)
                        const float2q ex = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                        const float2q d = ex + (float2q)(1.0f);
                        e[g] = (float2q){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    }
                    const float2q gg = __builtin_elementwise_fma(e[2], (float2q)(2.0f), (float2q)(-1.0f));
                    const float2q cprev = {cst[jj][rt][2 * h2], cst[jj][rt][2 * h2 + 1]};
                    float2q c = __builtin_elementwise_fma(e[1], cprev, e[0] * gg);
                    const float2q xc = c * (float2q)(-2.88539008f);
                    const float2q dc = (float2q){__builtin_amdgcn_exp2f(xc[0]), __builtin_amdgcn_exp2f(xc[1])} + (float2q)(1.0f);
                    const float2q rc = {__builtin_amdgcn_rcpf(dc[0]), __builtin_amdgcn_rcpf(dc[1])};
                    float2q hval = e[3] * __builtin_elementwise_fma(rc, (float2q)(2.0f), (float2q)(-1.0f));
                    if (MASKED && !((vm >> (rt * 16 + l15)) & 1ull)) {
                        c = (float2q)(0.0f);
                        hval = (float2q)(0.0f);
                    }
                    cst[jj][rt][2 * h2] = c[0];
                    cst[jj][rt][2 * h2 + 1] = c[1];
                    const float2q hq = hval * (float2q)(127.0f);
                    pk |= ((int)__builtin_rintf(hq[0]) & 0xff) << (16 * h2);
                    pk |= ((int)__builtin_rintf(hq[1]) & 0xff) << (16 * h2 + 8);
                    hv[2 * h2] = (half_t)hval[0];
                    hv[2 * h2 + 1] = (half_t)hval[1];
                }
                *(int *)(hnext + (rt * 16 + l15) * LDB + j * 16 + 4 * lq) = pk;
                if (OUT_F16) *(half4_t *)(hout + (rt * 16 + l15) * LDH + j * 16 + 4 * lq) = hv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 16), col = c % (C / 16);
            *(int4v *)(xbuf + row * LDB + col * 16) = xpf[p];
        }
        if (OUT_F16) {
            half_t *orow = (half_t *)Xout_ + ((size_t)t * N + n0) * C;
#pragma unroll
            for (int p = 0; p < 2 * XPF; ++p) {
                const int c = tid + NT * p;
                const int row = c / (C / 8), col8 = c % (C / 8);
                if (NTX) __builtin_nontemporal_store(*(const half8_t *)(hout + row * LDH + col8 * 8), (half8_t *)(orow + (size_t)c * 8));
                else *(half8_t *)(orow + (size_t)c * 8) = *(const half8_t *)(hout + row * LDH + col8 * 8);
            }
        } else {
            int8_t *orow = (int8_t *)Xout_ + ((size_t)t * N + n0) * C;
#pragma unroll
            for (int p = 0; p < XPF; ++p) {
                const int c = tid + NT * p;
                const int row = c / (C / 16), col = c % (C / 16);
                if (NTX) __builtin_nontemporal_store(*(const int4v *)(hnext + row * LDB + col * 16), (int4v *)(orow + (size_t)c * 16));
                else *(int4v *)(orow + (size_t)c * 16) = *(const int4v *)(hnext + row * LDB + col * 16);
            }
        }
        __syncthreads();
    }
}

// f16 -> int8 = round(127 clamp(v, -1, 1)) of the first (f16) layer's output: the reference's host_convert step
// (nn/LSTMStack.cpp:199-207).  HBM-bound: 3 B per element.
__global__ __launch_bounds__(256) void q8_convert_kernel(const half8_t *__restrict__ in, long long *__restrict__ out, size_t n8) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const half8_t v = in[i];
        unsigned long long pk = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = fminf(1.0f, fmaxf(-1.0f, (float)v[e]));
            pk |= (unsigned long long)(q8_quant(f) & 0xff) << (8 * e);
        }
        out[i] = (long long)pk;
    }
}

extern "C" int mibc_launch_q8_convert(hipStream_t s, const half_t *in, int8_t *out, size_t n) {
    if (n % 8 != 0) return 1;
    const size_t n8 = n / 8;
    size_t blocks = (n8 + 255) / 256;
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(q8_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const half8_t *)in, (long long *)out, n8);
    return 0;
}

template <int C, bool OUT_F16>
static size_t q8_lds_bytes() {
    return (size_t)3 * 64 * (C + 32) + (size_t)8 * C * 4 + (OUT_F16 ? (size_t)64 * (C + 16) * 2 : 0);
}

// int8 layer: Xin int8 [T][N][C]; Xout int8 [T][N][C] or (out_f16) f16 [T][N][C].  0 = launched, 1 = shape not covered.
// tmask != nullptr: the masked (variable chunk sizes) instances.
extern "C" int mibc_launch_lstm_layer_q8(hipStream_t s, int C, const int8_t *Xin, void *Xout, const int8_t *Wq,
                                         const float *biasn, const float *deqn, int T, int N, int reverse, int out_f16,
                                         const unsigned long long *tmask) {
    if (N % 64 != 0 || Wq == nullptr) return 1;
    dim3 grid(N / 64);
    static const int q8_nt = MIBC_ENV_INT("MIBC_Q8_NT", Q8_NT_DEFAULT);   // (debug build: A/B switch)
    static const int q8_stagger = MIBC_ENV_INT("MIBC_Q8_STAGGER", Q8_STAGGER_DEFAULT);
#define Q8N(CC, PF_, O_, M_, X_)                                                                                         \
    do {                                                                                                                 \
        MIBC_LDS_ATTR_ONCE((lstm_layer_q8_kernel<CC, PF_, O_, M_, X_>), (q8_lds_bytes<CC, O_>()));                       \
        hipLaunchKernelGGL((lstm_layer_q8_kernel<CC, PF_, O_, M_, X_>), grid, dim3(512), (q8_lds_bytes<CC, O_>()), s, Xin, Xout, Wq, \
                           biasn, deqn, T, N, reverse, tmask, q8_stagger);                                               \
        return 0;                                                                                                        \
    } while (0)
#define Q8(CC, PF_, O_, M_)                                                                                              \
    do {                                                                                                                 \
        if (q8_nt) Q8N(CC, PF_, O_, M_, true);                                                                           \
        else Q8N(CC, PF_, O_, M_, false);                                                                                \
    } while (0)
#define Q8M(CC)                                                  \
    do {                                                         \
        if (tmask) {                                             \
            if (out_f16) Q8(CC, 4, true, true);                  \
            else Q8(CC, 4, false, true);                         \
        }                                                        \
        if (out_f16) Q8(CC, 4, true, false);                     \
        else Q8(CC, 4, false, false);                            \
    } while (0)
    switch (C) {
        case 128: Q8M(128);
        case 256: Q8M(256);
        case 384: Q8M(384);
        default: return 1;
    }
#undef Q8M
#undef Q8
#undef Q8N
}
