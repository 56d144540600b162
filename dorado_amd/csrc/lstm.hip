// dorado_amd/csrc/lstm.hip — fused uni-directional LSTM layer.
//
// Replaces torch::nn::LSTM on the CPU path (dorado/nn/LSTMStack.cpp:19-41) and the Koi call
// host_cutlass_lstm (LSTMStack.cpp:193) on the CUDA path.  Semantics = torch LSTM: gates
// i,f,g,o = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh; c = s(f) c + s(i) tanh(g); h = s(o) tanh(c);
// zero initial state; `reverse` runs t = T-1 .. 0.
//
// One launch = one whole layer (all T steps inside the kernel; no per-step launches, no grid
// barrier): every workgroup owns NB = 32*RT chunks (batch rows) for all T steps, so the
// recurrence never leaves the CU.  Per step the workgroup computes the [4C x 2C] . [2C x NB]
// product of the concatenated operand [x_t ; h_{t-1}] on MFMA (v_mfma_f32_32x32x16_f16, fp32
// accumulate):
//   * weights are the MFMA A operand, streamed every step from L2 in a host-pre-tiled
//     "fragment order" ([hidden tile][k-step][gate][lane][8 halfs]) so that each load is one
//     fully coalesced 1 KiB wave transaction (buffer load, scalar offset: no address VALU);
//     a register ring keeps PF k-steps in flight continuously across gate epilogues, tiles and
//     time steps;
//   * h_{t-1} lives in LDS (f16, double-buffered, rows padded by 16 B -> conflict-free
//     ds_read_b128);
//   * the cell state c stays in registers (fp32) for the whole layer;
//   * the 4 gates of a hidden unit land in the same lane/register slot of 4 accumulators, so
//     the gate math is lane-local; h_t is written to LDS as packed 8-byte stores and leaves
//     for HBM as whole rows, 16 bytes per lane, after the step barrier.
// Two variants:
//   lstm_layer_xl_kernel<C, PF>      C <= 384, NB = 64, 4 waves: x_t is staged through LDS too
//                                    (fetched coalesced one step ahead into registers) — hac.
//   lstm_layer_xg_kernel<C, RT, NW>  any C (LDS permitting): x_t fragments straight from
//                                    global/L2 through their own ring — sup (C = 1024: NB = 32,
//                                    8 waves, 148 KB LDS) and C = 512.
// Wave w owns hidden tiles [w*HT, (w+1)*HT), HT = C/32/NW.
#include "common.h"

#define XG_PF 4

__device__ __forceinline__ half8_t wload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// acc[g][rt][*] <- bias of hidden tile j (LDS, natural hidden order [j][g][32])
template <int RT>
__device__ __forceinline__ void acc_init(float16_t (&acc)[4][RT], const float *bias_s, int j, int lhi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float *bp = bias_s + (j * 4 + g) * 32 + 4 * lhi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4_t v = *(const float4_t *)(bp + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[g][rt][q * 4 + e] = v[e];
        }
    }
}

// lane-local gate math for one hidden tile: D row = hidden unit, D col = batch row.
// MASKED (variable-chunk mode): bit (rowbit0 + rt*32 + l31) of vm = "this batch row is inside a chunk at this
// step"; outside, h and c are forced to 0 = the fresh initial state of the neighbouring chunk.
template <int RT, bool MASKED = false>
__device__ __forceinline__ void gates(const float16_t (&acc)[4][RT], float16_t (&cst)[RT], half_t *hnext,
                                      int LD, int j, int l31, int lhi, unsigned long long vm = ~0ull,
                                      int rowbit0 = 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        half_t *hdst = hnext + (rt * 32 + l31) * LD + j * 32 + 4 * lhi;
        const bool rowon = !MASKED || ((vm >> (rowbit0 + rt * 32 + l31)) & 1ull);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4_t hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = q * 4 + e;
                const float ig = fast_sigmoid(acc[0][rt][r]);
                const float fg = fast_sigmoid(acc[1][rt][r]);
                const float gg = fast_tanh(acc[2][rt][r]);
                const float og = fast_sigmoid(acc[3][rt][r]);
                float c = fmaf(fg, cst[rt][r], ig * gg);
                float hval = og * fast_tanh(c);
                if (MASKED && !rowon) {
                    c = 0.0f;
                    hval = 0.0f;
                }
                cst[rt][r] = c;
                hv[e] = (half_t)hval;
            }
            *(half4_t *)(hdst + 8 * q) = hv;
        }
    }
}

#ifdef MIBC_DEBUG_KERNELS   // superseded by x8 (kept for A/B timing in the debug build)
// ---------------------------------------------------------------------------------------------
// xl: x_t and h_{t-1} both in LDS (C <= 384)
// ---------------------------------------------------------------------------------------------
template <int C, int PF>
__global__ __launch_bounds__(256, 1) void lstm_layer_xl_kernel(
        const half_t *__restrict__ Xin,   // [T][N][C]
        half_t *__restrict__ Xout,        // [T][N][C]
        const half_t *__restrict__ Wf,    // [C/32][2C/16][4][64][8]
        const float *__restrict__ biasn,  // [C/32][4][32]  (b_ih + b_hh)
        int T, int N, int reverse) {
    constexpr int NB = 64;
    constexpr int HT = C / 128;
    constexpr int KS = 2 * C / 16;
    constexpr int KSX = C / 16;
    constexpr int LD = C + 8;
    constexpr int XPF = C / 32;  // 16-byte chunks per thread for one x_t block (64 rows)
    constexpr int KTOT = HT * KS;
    __shared__ __attribute__((aligned(16))) half_t hbuf[2][NB * LD];
    __shared__ __attribute__((aligned(16))) half_t xbuf[NB * LD];
    __shared__ __attribute__((aligned(16))) float bias_s[4 * C];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = blockIdx.x * NB;

    for (int i = tid; i < NB * LD / 8; i += 256) ((half8_t *)hbuf[0])[i] = (half8_t)(0);
    for (int i = tid; i < 4 * C; i += 256) bias_s[i] = biasn[i];

    float16_t cst[HT][2];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) cst[a][b][r] = 0.0f;

    // weights: one buffer resource per wave (its HT hidden tiles), scalar offset per k-step
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(Wf + (size_t)wave * HT * KS * 4 * 64 * 8), 0, HT * KS * 4 * 64 * 16, 0x00020000);
    const int wvoff = lane * 16;
    half8_t wr[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (u * 4 + g) * 1024);
    int kpre = PF;

    {  // x_{t0} -> LDS
        const int t_first = reverse ? (T - 1) : 0;
        const half_t *xg = Xin + ((size_t)t_first * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + 256 * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(xbuf + row * LD + col8 * 8) = *(const half8_t *)(xg + (size_t)c * 8);
        }
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? (T - 1 - step) : step;
        const int tn = (step + 1 < T) ? (reverse ? (t - 1) : (t + 1)) : t;
        const half_t *hprev = hbuf[step & 1];
        half_t *hnext = hbuf[(step + 1) & 1];

        // fetch x_{t+1} now, park it in registers, store it to LDS after this step's barrier
        half8_t xpf[XPF];
        {
            const half_t *xg = Xin + ((size_t)tn * N + n0) * C;
#pragma unroll
            for (int p = 0; p < XPF; ++p) xpf[p] = *(const half8_t *)(xg + (size_t)(tid + 256 * p) * 8);
        }

#pragma unroll
        for (int jj = 0; jj < HT; ++jj) {
            const int j = wave * HT + jj;
            float16_t acc[4][2];
            acc_init<2>(acc, bias_s, j, lhi);
#pragma unroll
            for (int phase = 0; phase < 2; ++phase) {
                const half_t *bsrc = (phase == 0 ? xbuf : hprev) + l31 * LD + 8 * lhi;
#pragma nounroll
                for (int ks0 = 0; ks0 < KSX; ks0 += PF) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const int ks = ks0 + u;
                        const half8_t b0 = *(const half8_t *)(bsrc + ks * 16);
                        const half8_t b1 = *(const half8_t *)(bsrc + 32 * LD + ks * 16);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc[g][0] = mfma32x32x16(wr[u][g], b0, acc[g][0]);
                            acc[g][1] = mfma32x32x16(wr[u][g], b1, acc[g][1]);
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (kpre * 4 + g) * 1024);
                        kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                    }
                }
            }
            gates<2>(acc, cst[jj], hnext, LD, j, l31, lhi);
        }
        __syncthreads();  // h_t complete; nobody reads x_t / h_{t-1} any more
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + 256 * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(xbuf + row * LD + col8 * 8) = xpf[p];
        }
        half_t *orow = Xout + ((size_t)t * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + 256 * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(orow + (size_t)c * 8) = *(const half8_t *)(hnext + row * LD + col8 * 8);
        }
        __syncthreads();  // x_{t+1} visible
    }
}

#endif  // MIBC_DEBUG_KERNELS

// ---------------------------------------------------------------------------------------------
// xg: x_t fragments straight from global/L2 (any C); NB = 32*RT rows, NW waves
// ---------------------------------------------------------------------------------------------
template <int C, int RT, int NW, int PF, bool MASKED = false>
__global__ __launch_bounds__(64 * NW, 1) void lstm_layer_xg_kernel(
        const half_t *__restrict__ Xin, half_t *__restrict__ Xout, const half_t *__restrict__ Wf,
        const float *__restrict__ biasn, int T, int N, int reverse,
        const unsigned long long *__restrict__ tmask = nullptr /* MASKED: [T][N/64], see x8 */) {
    constexpr int NB = 32 * RT;
    constexpr int NT = 64 * NW;
    constexpr int HT = C / 32 / NW;
    constexpr int KS = 2 * C / 16;
    constexpr int KSX = C / 16;
    constexpr int LD = C + 8;
    constexpr int KTOT = HT * KS;
    static_assert((C / 16) % PF == 0, "ring depth must divide the k-steps of each part");
    __shared__ __attribute__((aligned(16))) half_t hbuf[2][NB * LD];
    __shared__ __attribute__((aligned(16))) float bias_s[4 * C];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = blockIdx.x * NB;

    for (int i = tid; i < NB * LD / 8; i += NT) ((half8_t *)hbuf[0])[i] = (half8_t)(0);
    for (int i = tid; i < 4 * C; i += NT) bias_s[i] = biasn[i];

    float16_t cst[HT][RT];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) cst[a][b][r] = 0.0f;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(Wf + (size_t)wave * HT * KS * 4 * 64 * 8), 0, HT * KS * 4 * 64 * 16, 0x00020000);
    const int wvoff = lane * 16;
    half8_t wr[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (u * 4 + g) * 1024);
    int kpre = PF;

    half8_t xr[PF][RT];
    {
        const int t_first = reverse ? (T - 1) : 0;
        const half_t *xp0 = Xin + ((size_t)t_first * N + n0 + l31) * C + 8 * lhi;
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) xr[u][rt] = *(const half8_t *)(xp0 + (size_t)rt * 32 * C + u * 16);
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? (T - 1 - step) : step;
        const int tn = (step + 1 < T) ? (reverse ? (t - 1) : (t + 1)) : t;
        const half_t *xp_cur = Xin + ((size_t)t * N + n0 + l31) * C + 8 * lhi;
        const half_t *xp_step_next = Xin + ((size_t)tn * N + n0 + l31) * C + 8 * lhi;
        const half_t *hprev = hbuf[step & 1];
        half_t *hnext = hbuf[(step + 1) & 1];
        unsigned long long vm = ~0ull;
        if (MASKED) vm = tmask[(size_t)t * (N / 64) + (n0 >> 6)];

#pragma unroll
        for (int jj = 0; jj < HT; ++jj) {
            const int j = wave * HT + jj;
            float16_t acc[4][RT];
            acc_init<RT>(acc, bias_s, j, lhi);
            // x part: its ring also runs continuously (wraps into the next tile / next step)
            const half_t *xp_next = (jj == HT - 1) ? xp_step_next : xp_cur;
#pragma nounroll
            for (int ks0 = 0; ks0 < KSX; ks0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int ks = ks0 + u;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[g][rt] = mfma32x32x16(wr[u][g], xr[u][rt], acc[g][rt]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (kpre * 4 + g) * 1024);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                    const int kn = ks + PF;
                    const half_t *xq = (kn < KSX) ? (xp_cur + kn * 16) : (xp_next + (kn - KSX) * 16);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) xr[u][rt] = *(const half8_t *)(xq + (size_t)rt * 32 * C);
                }
            }
            // h part from LDS
            const half_t *hp = hprev + l31 * LD + 8 * lhi;
#pragma nounroll
            for (int ks0 = 0; ks0 < KSX; ks0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int ks = ks0 + u;
                    half8_t hb[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) hb[rt] = *(const half8_t *)(hp + rt * 32 * LD + ks * 16);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[g][rt] = mfma32x32x16(wr[u][g], hb[rt], acc[g][rt]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (kpre * 4 + g) * 1024);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                }
            }
            gates<RT, MASKED>(acc, cst[jj], hnext, LD, j, l31, lhi, vm, n0 & 63);
        }
        __syncthreads();
        half_t *orow = Xout + ((size_t)t * N + n0) * C;
        for (int i = tid; i < NB * (C / 8); i += NT) {
            const int row = i / (C / 8), seg = i % (C / 8);
            *(half8_t *)(orow + (size_t)row * C + seg * 8) = *(const half8_t *)(hnext + row * LD + seg * 8);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// x8: like xl (x_t and h_{t-1} in LDS, NB = 64) but 8 waves = 2 per SIMD, each owning C/128
// hidden tiles of 16 units on v_mfma_f32_16x16x32_f16: per-wave register footprint drops under
// 256, so two waves share a SIMD and one wave's gate epilogue / LDS waits / barrier skew are
// covered by the other wave's MFMAs.  Weight traffic is unchanged (every wave streams only its
// own hidden units' rows).  D layout of 16x16x32: col = lane & 15, row = 4*(lane >> 4) + reg;
// A/B: lane holds [i = lane & 15][k = 8*(lane >> 4) .. +8].
// ---------------------------------------------------------------------------------------------
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4v mfma16x16x32(half8_t a, half8_t b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int C, int PF, bool MASKED = false>
__global__ __launch_bounds__(512, 2) void lstm_layer_x8_kernel(
        const half_t *__restrict__ Xin,   // [T][N][C]
        half_t *__restrict__ Xout,        // [T][N][C]
        const half_t *__restrict__ Wf16,  // [C/16][2C/32][4][64][8]
        const float *__restrict__ biasn,  // [4C]: [(hidden/32)][4][32]  (b_ih + b_hh)
        int T, int N, int reverse,
        // variable-chunk mode (MASKED): bit r of tmask[t * gridDim.x + blockIdx.x] = "row r of this workgroup is
        // inside a chunk at step t"; outside (the >= 2 gap steps between packed chunks) h and c are forced to 0,
        // which is the zero initial state of the next chunk in either direction (nn/LSTMStack.cpp:29-41)
        const unsigned long long *__restrict__ tmask = nullptr) {
    constexpr int NB = 64;
    constexpr int NT = 512;
    constexpr int HT = C / 16 / 8;   // 16-unit hidden tiles per wave
    constexpr int KS = 2 * C / 32;   // k-steps of 32 over [x ; h]
    constexpr int KSX = C / 32;
    constexpr int LD = C + 16;       // +32 B: conflict-free ds_read_b128 for the 16x32 fragment pattern
    constexpr int XPF = C / 64;      // 16-byte chunks per thread for one x_t block
    constexpr int KTOT = HT * KS;
    constexpr int UN = (PF & 1) ? 2 * PF : PF;  // unroll: ring slot u % PF, fragment parity u & 1
    static_assert(KS % UN == 0, "k-steps must divide the unroll");
    __shared__ __attribute__((aligned(16))) half_t hbuf[2][NB * LD];
    __shared__ __attribute__((aligned(16))) half_t xbuf[NB * LD];
    __shared__ __attribute__((aligned(16))) float bias_s[4 * C];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * NB;

    for (int i = tid; i < NB * LD / 8; i += NT) ((half8_t *)hbuf[0])[i] = (half8_t)(0);
    for (int i = tid; i < 4 * C; i += NT) bias_s[i] = biasn[i];

    float4v cst[HT][4];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cst[a][b] = (float4v)(0.0f);

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(Wf16 + (size_t)wave * HT * KS * 4 * 64 * 8), 0, HT * KS * 4 * 64 * 16, 0x00020000);
    const int wvoff = lane * 16;
    half8_t wr[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (u * 4 + g) * 1024);
    int kpre = PF;

    {
        const int t_first = reverse ? (T - 1) : 0;
        const half_t *xg = Xin + ((size_t)t_first * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(xbuf + row * LD + col8 * 8) = *(const half8_t *)(xg + (size_t)c * 8);
        }
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? (T - 1 - step) : step;
        const int tn = (step + 1 < T) ? (reverse ? (t - 1) : (t + 1)) : t;
        const half_t *hprev = hbuf[step & 1];
        half_t *hnext = hbuf[(step + 1) & 1];

        unsigned long long vm = ~0ull;
        if (MASKED) vm = tmask[(size_t)t * gridDim.x + blockIdx.x];
        half8_t xpf[XPF];
        {
            // non-temporal: x_t is read once and h_t (below) is read by the NEXT launch — kept out of the way, the layer's 21 GB
            // activation streams no longer push the 2.36 MB weight set out of the XCD's L2 (round 5, same-box A/B over five
            // alternations: LSTM stack 297.0 -> 293.3 ms, profiles/r05_h_x8_nt_ab.log); cache policy only, results unchanged
            const half_t *xg = Xin + ((size_t)tn * N + n0) * C;
#pragma unroll
            for (int p = 0; p < XPF; ++p) xpf[p] = __builtin_nontemporal_load((const half8_t *)(xg + (size_t)(tid + NT * p) * 8));
        }

#pragma unroll
        for (int jj = 0; jj < HT; ++jj) {
            const int j = wave * HT + jj;  // 16-unit hidden tile
            float4v acc[4][4];             // [gate][row tile of 16]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // hidden = 16 j + 4 lq + r ; bias_s order [(hidden / 32)][g][hidden % 32]
                const float4v bv = *(const float4v *)(bias_s + ((j >> 1) * 4 + g) * 32 + (j & 1) * 16 + 4 * lq);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[g][rt] = bv;
            }
            const half_t *xb = xbuf + l15 * LD + 8 * lq;
            const half_t *hb = hprev + l15 * LD + 8 * lq;
            half8_t bq[2][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) bq[0][rt] = *(const half8_t *)(xb + rt * 16 * LD);
#pragma nounroll
            for (int ks0 = 0; ks0 < KS; ks0 += UN) {
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) {
                    const int u = uu % PF;
                    const int kn = (ks0 + uu + 1 < KS) ? (ks0 + uu + 1) : (KS - 1);
                    const half_t *bn = (kn < KSX) ? (xb + kn * 32) : (hb + (kn - KSX) * 32);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) bq[(uu + 1) & 1][rt] = *(const half8_t *)(bn + rt * 16 * LD);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt)
                            acc[g][rt] = mfma16x16x32(wr[u][g], bq[uu & 1][rt], acc[g][rt]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (kpre * 4 + g) * 1024);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                }
            }
            // gates: D row = hidden 4 lq + r, D col = batch row l15 of row tile rt
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                half4_t hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ig = fast_sigmoid(acc[0][rt][r]);
                    const float fg = fast_sigmoid(acc[1][rt][r]);
                    const float gg = fast_tanh(acc[2][rt][r]);
                    const float og = fast_sigmoid(acc[3][rt][r]);
                    float c = fmaf(fg, cst[jj][rt][r], ig * gg);
                    float hval = og * fast_tanh(c);
                    if (MASKED && !((vm >> (rt * 16 + l15)) & 1ull)) {
                        c = 0.0f;
                        hval = 0.0f;
                    }
                    cst[jj][rt][r] = c;
                    hv[r] = (half_t)hval;
                }
                *(half4_t *)(hnext + (rt * 16 + l15) * LD + j * 16 + 4 * lq) = hv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(xbuf + row * LD + col8 * 8) = xpf[p];
        }
        half_t *orow = Xout + ((size_t)t * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            __builtin_nontemporal_store(*(const half8_t *)(hnext + row * LD + col8 * 8), (half8_t *)(orow + (size_t)c * 8));
        }
        __syncthreads();
    }
}

#ifdef MIBC_DEBUG_KERNELS
// Ablation copy of x8 for timing experiments (MIBC_LSTM_DBG=<bits>; results are wrong when a bit is
// set): bit0 = no gate math, bit1 = no weight reloads, bit2 = no activation fragment reads;
// 8 = unmodified.  DESIGN.md "what bounds the LSTM kernel" quotes these runs.
template <int C, int PF, int DBG>
__global__ __launch_bounds__(512, 2) void lstm_layer_x8dbg_kernel(
        const half_t *__restrict__ Xin,   // [T][N][C]
        half_t *__restrict__ Xout,        // [T][N][C]
        const half_t *__restrict__ Wf16,  // [C/16][2C/32][4][64][8]
        const float *__restrict__ biasn,  // [4C]: [(hidden/32)][4][32]  (b_ih + b_hh)
        int T, int N, int reverse) {
    constexpr int NB = 64;
    constexpr int NT = 512;
    constexpr int HT = C / 16 / 8;   // 16-unit hidden tiles per wave
    constexpr int KS = 2 * C / 32;   // k-steps of 32 over [x ; h]
    constexpr int KSX = C / 32;
    constexpr int LD = C + 16;       // +32 B: conflict-free ds_read_b128 for the 16x32 fragment pattern
    constexpr int XPF = C / 64;      // 16-byte chunks per thread for one x_t block
    constexpr int KTOT = HT * KS;
    constexpr int UN = (PF & 1) ? 2 * PF : PF;  // unroll: ring slot u % PF, fragment parity u & 1
    static_assert(KS % UN == 0, "k-steps must divide the unroll");
    __shared__ __attribute__((aligned(16))) half_t hbuf[2][NB * LD];
    __shared__ __attribute__((aligned(16))) half_t xbuf[NB * LD];
    __shared__ __attribute__((aligned(16))) float bias_s[4 * C];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * NB;

    for (int i = tid; i < NB * LD / 8; i += NT) ((half8_t *)hbuf[0])[i] = (half8_t)(0);
    for (int i = tid; i < 4 * C; i += NT) bias_s[i] = biasn[i];

    float4v cst[HT][4];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cst[a][b] = (float4v)(0.0f);

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(Wf16 + (size_t)wave * HT * KS * 4 * 64 * 8), 0, HT * KS * 4 * 64 * 16, 0x00020000);
    const int wvoff = lane * 16;
    half8_t wr[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[u][g] = wload(wrs, wvoff, (u * 4 + g) * 1024);
    int kpre = PF;

    {
        const int t_first = reverse ? (T - 1) : 0;
        const half_t *xg = Xin + ((size_t)t_first * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(xbuf + row * LD + col8 * 8) = *(const half8_t *)(xg + (size_t)c * 8);
        }
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? (T - 1 - step) : step;
        const int tn = (step + 1 < T) ? (reverse ? (t - 1) : (t + 1)) : t;
        const half_t *hprev = hbuf[step & 1];
        half_t *hnext = hbuf[(step + 1) & 1];

        half8_t xpf[XPF];
        {
            const half_t *xg = Xin + ((size_t)tn * N + n0) * C;
#pragma unroll
            for (int p = 0; p < XPF; ++p)
                xpf[p] = (DBG & 16) ? __builtin_nontemporal_load((const half8_t *)(xg + (size_t)(tid + NT * p) * 8))   // round 5 A/B: x_t is read once
                                    : *(const half8_t *)(xg + (size_t)(tid + NT * p) * 8);
        }

#pragma unroll
        for (int jj = 0; jj < HT; ++jj) {
            const int j = wave * HT + jj;  // 16-unit hidden tile
            float4v acc[4][4];             // [gate][row tile of 16]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // hidden = 16 j + 4 lq + r ; bias_s order [(hidden / 32)][g][hidden % 32]
                const float4v bv = *(const float4v *)(bias_s + ((j >> 1) * 4 + g) * 32 + (j & 1) * 16 + 4 * lq);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[g][rt] = bv;
            }
            const half_t *xb = xbuf + l15 * LD + 8 * lq;
            const half_t *hb = hprev + l15 * LD + 8 * lq;
            half8_t bq[2][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) bq[0][rt] = *(const half8_t *)(xb + rt * 16 * LD);
#pragma nounroll
            for (int ks0 = 0; ks0 < KS; ks0 += UN) {
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) {
                    const int u = uu % PF;
                    const int kn = (ks0 + uu + 1 < KS) ? (ks0 + uu + 1) : (KS - 1);
                    const half_t *bn = (kn < KSX) ? (xb + kn * 32) : (hb + (kn - KSX) * 32);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
                        if (!(DBG & 4)) bq[(uu + 1) & 1][rt] = *(const half8_t *)(bn + rt * 16 * LD);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt)
                            acc[g][rt] = mfma16x16x32(wr[u][g], bq[uu & 1][rt], acc[g][rt]);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (!(DBG & 2)) wr[u][g] = wload(wrs, wvoff, (kpre * 4 + g) * 1024);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                }
            }
            // gates: D row = hidden 4 lq + r, D col = batch row l15 of row tile rt
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                half4_t hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (DBG & 1) {
                        hv[r] = (half_t)(1e-3f * (acc[0][rt][r] + acc[1][rt][r] + acc[2][rt][r] + acc[3][rt][r]));
                    } else {
                        const float ig = fast_sigmoid(acc[0][rt][r]);
                        const float fg = fast_sigmoid(acc[1][rt][r]);
                        const float gg = fast_tanh(acc[2][rt][r]);
                        const float og = fast_sigmoid(acc[3][rt][r]);
                        const float c = fmaf(fg, cst[jj][rt][r], ig * gg);
                        cst[jj][rt][r] = c;
                        hv[r] = (half_t)(og * fast_tanh(c));
                    }
                }
                *(half4_t *)(hnext + (rt * 16 + l15) * LD + j * 16 + 4 * lq) = hv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            *(half8_t *)(xbuf + row * LD + col8 * 8) = xpf[p];
        }
        half_t *orow = Xout + ((size_t)t * N + n0) * C;
#pragma unroll
        for (int p = 0; p < XPF; ++p) {
            const int c = tid + NT * p;
            const int row = c / (C / 8), col8 = c % (C / 8);
            if (DBG & 32) __builtin_nontemporal_store(*(const half8_t *)(hnext + row * LD + col8 * 8), (half8_t *)(orow + (size_t)c * 8));   // round 5 A/B
            else *(half8_t *)(orow + (size_t)c * 8) = *(const half8_t *)(hnext + row * LD + col8 * 8);
        }
        __syncthreads();
    }
}

#endif  // MIBC_DEBUG_KERNELS

// batch granularity (rows per workgroup) for a given layer width
extern "C" int mibc_lstm_rows_per_wg(int C) {
    if (C == 96 || C == 128 || C == 256 || C == 384 || C == 512) return 64;
    if (C == 768 || C == 1024) return 32;
    return 0;
}

// Variable-chunk mode: masked x8 kernels (C = 128 / 256 / 384) and masked xg kernels (C = 512 / 768 / 1024 —
// the widths for which the reference enables variable chunk sizes, api/runner_creation.cpp:24-44).
extern "C" int mibc_launch_lstm_layer_masked(hipStream_t s, int C, const half_t *Xin, half_t *Xout,
                                             const half_t *Wf16, const float *biasn, int T, int N, int reverse,
                                             const unsigned long long *tmask) {
    if (tmask == nullptr || N % 64 != 0) return 1;
    if (C >= 512) {
        // Wf16 carries the 32-unit-tile layout (lstm_w) for these widths
        if (Wf16 == nullptr) return 1;
        const int nb = mibc_lstm_rows_per_wg(C);
        dim3 grid(N / nb);
#define XGM(CC, RT, NW) hipLaunchKernelGGL((lstm_layer_xg_kernel<CC, RT, NW, 4, true>), grid, dim3(64 * NW), 0, s, Xin, Xout, Wf16, biasn, T, N, reverse, tmask)
        switch (C) {
            case 512: XGM(512, 2, 4); return 0;
            case 768: XGM(768, 1, 8); return 0;
            case 1024: XGM(1024, 1, 8); return 0;
            default: return 1;
        }
#undef XGM
    }
    if (Wf16 == nullptr) return 1;
    dim3 g8(N / 64);
    switch (C) {
        case 128: hipLaunchKernelGGL((lstm_layer_x8_kernel<128, 4, true>), g8, dim3(512), 0, s, Xin, Xout, Wf16, biasn, T, N, reverse, tmask); return 0;
        case 256: hipLaunchKernelGGL((lstm_layer_x8_kernel<256, 4, true>), g8, dim3(512), 0, s, Xin, Xout, Wf16, biasn, T, N, reverse, tmask); return 0;
        case 384: hipLaunchKernelGGL((lstm_layer_x8_kernel<384, 4, true>), g8, dim3(512), 0, s, Xin, Xout, Wf16, biasn, T, N, reverse, tmask); return 0;
        default: return 1;
    }
}

extern "C" int mibc_launch_lstm_layer(hipStream_t s, int C, const half_t *Xin, half_t *Xout,
                                      const half_t *Wf, const half_t *Wf16, const float *biasn, int T, int N,
                                      int reverse) {
    const int nb = mibc_lstm_rows_per_wg(C);
    if (nb == 0 || N % nb != 0) {
        return 1;
    }
    dim3 grid(N / nb);
#define XG(CC, RT, NW) hipLaunchKernelGGL((lstm_layer_xg_kernel<CC, RT, NW, ((CC / 16) % 4 == 0 ? 4 : 2)>), grid, dim3(64 * NW), 0, s, Xin, Xout, Wf, biasn, T, N, reverse)
#define X8(CC) hipLaunchKernelGGL((lstm_layer_x8_kernel<CC, 4>), grid, dim3(512), 0, s, Xin, Xout, Wf16, biasn, T, N, reverse, (const unsigned long long *)nullptr)
#ifdef MIBC_DEBUG_KERNELS
    static const int dbg8 = MIBC_ENV_INT("MIBC_LSTM_DBG", 0);
    if (dbg8 && C == 384 && Wf16 != nullptr) {
#define X8D(D) case D: hipLaunchKernelGGL((lstm_layer_x8dbg_kernel<384, 4, D>), grid, dim3(512), 0, s, Xin, Xout, Wf16, biasn, T, N, reverse); return 0;
        switch (dbg8) { X8D(1) X8D(2) X8D(4) X8D(7) X8D(8) X8D(24) X8D(40) X8D(56) default: break; }
#undef X8D
    }
    static const int use_x8 = MIBC_ENV_INT("MIBC_LSTM_X8", 1);
    static const int force_xg = MIBC_ENV_INT("MIBC_LSTM_XG", 0);
    if (!use_x8 && (C == 128 || C == 256 || C == 384)) {
#define XL(CC, PF) hipLaunchKernelGGL((lstm_layer_xl_kernel<CC, PF>), grid, dim3(256), 0, s, Xin, Xout, Wf, biasn, T, N, reverse)
        switch (C) {
            case 128: if (force_xg) XG(128, 2, 4); else XL(128, 8); return 0;
            case 256: if (force_xg) XG(256, 2, 4); else XL(256, 8); return 0;
            default: if (force_xg) XG(384, 2, 4); else XL(384, 4); return 0;
        }
#undef XL
    }
#endif
    switch (C) {
        case 96: XG(96, 2, 3); return 0;   // fast models: 3 hidden tiles of 32 -> 3 waves
        case 128: if (Wf16 == nullptr) return 1; X8(128); return 0;
        case 256: if (Wf16 == nullptr) return 1; X8(256); return 0;
        case 384: if (Wf16 == nullptr) return 1; X8(384); return 0;
        case 512: XG(512, 2, 4); return 0;
        case 768: XG(768, 1, 8); return 0;
        case 1024: XG(1024, 1, 8); return 0;
        default: return 1;
    }
#undef X8
#undef XG
}
