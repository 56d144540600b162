// dorado_amd/csrc/lstm.hip — fused uni-directional LSTM layer.
//
// Replaces torch::nn::LSTM on the CPU path (dorado/nn/LSTMStack.cpp:19-41) and the Koi call
// host_cutlass_lstm (LSTMStack.cpp:193) on the CUDA path.  Semantics = torch LSTM: gates
// i,f,g,o = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh; c = s(f) c + s(i) tanh(g); h = s(o) tanh(c);
// zero initial state; `reverse` runs t = T-1 .. 0.
//
// One launch = one whole layer (all T steps inside the kernel; no per-step launches, no grid
// barrier): every workgroup owns NB = 64 chunks (batch rows) for all T steps, so the recurrence
// never leaves the CU.  Per step the workgroup computes the [4C x 2C] . [2C x NB] product of the
// concatenated operand [x_t ; h_{t-1}] on MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate):
//   * weights are the MFMA A operand, streamed every step from L2 in a host-pre-tiled
//     "fragment order" ([hidden tile][k-step][gate][lane][8 halfs]) so that each load is one
//     fully coalesced 1 KiB wave transaction; a 4-deep register ring keeps them in flight
//     continuously across gate epilogues, tiles and time steps;
//   * h_{t-1} lives in LDS (f16, double-buffered, rows padded by 16 B -> conflict-free
//     ds_read_b128), x_t is read straight from HBM/L2 (time-major [T][N][C] layout makes the
//     workgroup's x_t block one contiguous 48 KiB span);
//   * the cell state c stays in registers (fp32) for the whole layer;
//   * the 4 gates of a hidden unit land in the same lane/register slot of 4 accumulators, so
//     the gate math is lane-local; h_t is written to LDS as packed 8-byte stores and leaves
//     for HBM as whole 16-byte-per-lane rows after the step barrier.
// Wave w owns hidden tiles [w*HT, (w+1)*HT), HT = C/128 (hac: 3, sup: 8).
#include "common.h"
#include <stdlib.h>

#define L_NB 64
#define L_PF 4  // weight ring depth (k-steps in flight)

template <int C>
__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(
        const half_t *__restrict__ Xin,   // [T][N][C]
        half_t *__restrict__ Xout,        // [T][N][C]
        const half_t *__restrict__ Wf,    // [C/32][2C/16][4][64][8]
        const float *__restrict__ biasf,  // [C/32][4][2][16]   (b_ih + b_hh in D-register order)
        int T, int N, int reverse) {
    constexpr int HT = C / 128;      // hidden tiles per wave
    constexpr int KS = 2 * C / 16;   // k-steps over [x ; h]
    constexpr int KSX = C / 16;      // k-steps of the x part
    constexpr int LD = C + 8;        // LDS row stride (halfs)
    __shared__ __attribute__((aligned(16))) half_t hbuf[2][L_NB * LD];
    // biases in LDS (read at every tile start; an LDS read cannot be hoisted out of the time
    // loop across the barriers, a global read of loop-invariant data would be — 64 VGPRs/tile)
    __shared__ __attribute__((aligned(16))) float bias_s[4 * C * 2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = blockIdx.x * L_NB;

    for (int i = tid; i < L_NB * LD / 8; i += 256) {
        ((half8_t *)hbuf[0])[i] = (half8_t)(0);
    }
    for (int i = tid; i < 4 * C * 2; i += 256) {
        bias_s[i] = biasf[i];
    }

    float16_t cst[HT][2];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) cst[a][b][r] = 0.0f;

    // weight ring: runs continuously over kidx = (jj, ks) of this wave and wraps every step
    const half_t *wbase = Wf + ((size_t)(wave * HT) * KS * 4 * 64 + lane) * 8;
    constexpr int WSTRIDE = 4 * 64 * 8;  // halfs per k-step
    constexpr int KTOT = HT * KS;
    half8_t wr[L_PF][4];
#pragma unroll
    for (int u = 0; u < L_PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[u][g] = *(const half8_t *)(wbase + (size_t)u * WSTRIDE + g * 512);
    int kpre = L_PF;  // next kidx to prefetch (mod KTOT)
    half8_t xr[L_PF][2];
    {
        const int t_first = reverse ? (T - 1) : 0;
        const half_t *xp0 = Xin + ((size_t)t_first * N + n0 + l31) * C + 8 * lhi;
#pragma unroll
        for (int u = 0; u < L_PF; ++u) {
            xr[u][0] = *(const half8_t *)(xp0 + u * 16);
            xr[u][1] = *(const half8_t *)(xp0 + (size_t)32 * C + u * 16);
        }
    }

    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? (T - 1 - step) : step;
        const int tn = (step + 1 < T) ? (reverse ? (t - 1) : (t + 1)) : t;
        const half_t *xp_cur = Xin + ((size_t)t * N + n0 + l31) * C + 8 * lhi;
        const half_t *xp_step_next = Xin + ((size_t)tn * N + n0 + l31) * C + 8 * lhi;
        const half_t *hprev = hbuf[step & 1];
        half_t *hnext = hbuf[(step + 1) & 1];

#pragma unroll
        for (int jj = 0; jj < HT; ++jj) {
            const int j = wave * HT + jj;
            float16_t acc[4][2];
            {
                const float *bp = bias_s + (j * 4 * 2 + lhi) * 16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4_t *b4 = (const float4_t *)(bp + g * 32);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4_t v = b4[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[g][0][q * 4 + e] = v[e];
                            acc[g][1][q * 4 + e] = v[e];
                        }
                    }
                }
            }
            // ---- x part: B operand straight from global, through its own 4-deep ring that
            //      also runs continuously (wraps into the next tile / next time step) ----
            const half_t *xp_next = (jj == HT - 1) ? xp_step_next : xp_cur;
#pragma nounroll
            for (int ks0 = 0; ks0 < KSX; ks0 += L_PF) {
#pragma unroll
                for (int u = 0; u < L_PF; ++u) {
                    const int ks = ks0 + u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g][0] = mfma32x32x16(wr[u][g], xr[u][0], acc[g][0]);
                        acc[g][1] = mfma32x32x16(wr[u][g], xr[u][1], acc[g][1]);
                    }
                    const half_t *wp = wbase + (size_t)kpre * WSTRIDE;
#pragma unroll
                    for (int g = 0; g < 4; ++g) wr[u][g] = *(const half8_t *)(wp + g * 512);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                    const int kn = ks + L_PF;
                    const half_t *xq = (kn < KSX) ? (xp_cur + kn * 16) : (xp_next + (kn - KSX) * 16);
                    xr[u][0] = *(const half8_t *)(xq);
                    xr[u][1] = *(const half8_t *)(xq + (size_t)32 * C);
                }
            }
            // ---- h part: B operand from LDS ----
            const half_t *hp = hprev + l31 * LD + 8 * lhi;
#pragma nounroll
            for (int ks0 = 0; ks0 < KSX; ks0 += L_PF) {
#pragma unroll
                for (int u = 0; u < L_PF; ++u) {
                    const int ks = ks0 + u;
                    half8_t hb0 = *(const half8_t *)(hp + ks * 16);
                    half8_t hb1 = *(const half8_t *)(hp + 32 * LD + ks * 16);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g][0] = mfma32x32x16(wr[u][g], hb0, acc[g][0]);
                        acc[g][1] = mfma32x32x16(wr[u][g], hb1, acc[g][1]);
                    }
                    const half_t *wp = wbase + (size_t)kpre * WSTRIDE;
#pragma unroll
                    for (int g = 0; g < 4; ++g) wr[u][g] = *(const half8_t *)(wp + g * 512);
                    kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                }
            }
            // ---- gates (lane-local): D row = hidden unit, D col = batch row ----
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                half_t *hdst = hnext + (nb * 32 + l31) * LD + j * 32 + 4 * lhi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4_t hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = q * 4 + e;
                        const float ig = fast_sigmoid(acc[0][nb][r]);
                        const float fg = fast_sigmoid(acc[1][nb][r]);
                        const float gg = fast_tanh(acc[2][nb][r]);
                        const float og = fast_sigmoid(acc[3][nb][r]);
                        const float c = fmaf(fg, cst[jj][nb][r], ig * gg);
                        cst[jj][nb][r] = c;
                        hv[e] = (half_t)(og * fast_tanh(c));
                    }
                    *(half4_t *)(hdst + 8 * q) = hv;
                }
            }
        }
        __syncthreads();
        // h_t -> HBM, whole rows, 16 B per lane
        half_t *orow = Xout + ((size_t)t * N + n0) * C;
        for (int i = tid; i < L_NB * (C / 8); i += 256) {
            const int row = i / (C / 8), seg = i % (C / 8);
            *(half8_t *)(orow + (size_t)row * C + seg * 8) =
                    *(const half8_t *)(hnext + row * LD + seg * 8);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// v2: x_t staged through LDS (each row read from HBM once per workgroup, fully coalesced, fetched
// one step ahead into registers), weights through buffer loads with scalar offsets (no per-load
// VALU address math) and a deeper ring.  Used for C <= 384 (LDS: 2 h buffers + 1 x buffer).
// ---------------------------------------------------------------------------------------------

template <int C, int PF>
__global__ __launch_bounds__(256, 1) void lstm_layer_v2_kernel(
        const half_t *__restrict__ Xin,   // [T][N][C]
        half_t *__restrict__ Xout,        // [T][N][C]
        const half_t *__restrict__ Wf,    // [C/32][2C/16][4][64][8]
        const float *__restrict__ biasn,  // [C/32][4][32]  (b_ih + b_hh), hidden-unit order
        int T, int N, int reverse) {
    constexpr int HT = C / 128;
    constexpr int KS = 2 * C / 16;
    constexpr int KSX = C / 16;
    constexpr int LD = C + 8;
    constexpr int XPF = C / 32;  // 16-byte chunks per thread for one x_t block (64 rows)
    constexpr int KTOT = HT * KS;
    __shared__ __attribute__((aligned(16))) half_t hbuf[2][L_NB * LD];
    __shared__ __attribute__((aligned(16))) half_t xbuf[L_NB * LD];
    __shared__ __attribute__((aligned(16))) float bias_s[4 * C];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = blockIdx.x * L_NB;

    for (int i = tid; i < L_NB * LD / 8; i += 256) ((half8_t *)hbuf[0])[i] = (half8_t)(0);
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
        for (int g = 0; g < 4; ++g)
            wr[u][g] = __builtin_bit_cast(
                    half8_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, (u * 4 + g) * 1024, 0));
    int kpre = PF;

    // x_{t0} -> LDS
    {
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
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float *bp = bias_s + (j * 4 + g) * 32 + 4 * lhi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4_t v = *(const float4_t *)(bp + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[g][0][q * 4 + e] = v[e];
                        acc[g][1][q * 4 + e] = v[e];
                    }
                }
            }
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
                        for (int g = 0; g < 4; ++g)
                            wr[u][g] = __builtin_bit_cast(
                                    half8_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                     wrs, wvoff, (kpre * 4 + g) * 1024, 0));
                        kpre = (kpre + 1 == KTOT) ? 0 : kpre + 1;
                    }
                }
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                half_t *hdst = hnext + (nb * 32 + l31) * LD + j * 32 + 4 * lhi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4_t hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = q * 4 + e;
                        const float ig = fast_sigmoid(acc[0][nb][r]);
                        const float fg = fast_sigmoid(acc[1][nb][r]);
                        const float gg = fast_tanh(acc[2][nb][r]);
                        const float og = fast_sigmoid(acc[3][nb][r]);
                        const float c = fmaf(fg, cst[jj][nb][r], ig * gg);
                        cst[jj][nb][r] = c;
                        hv[e] = (half_t)(og * fast_tanh(c));
                    }
                    *(half4_t *)(hdst + 8 * q) = hv;
                }
            }
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

extern "C" int mibc_launch_lstm_layer(hipStream_t s, int C, const half_t *Xin, half_t *Xout,
                                      const half_t *Wf, const float *biasf, const float *biasn,
                                      int T, int N, int reverse) {
    static const int ver = getenv("MIBC_LSTM_V") ? atoi(getenv("MIBC_LSTM_V")) : 2;
    if (N % L_NB != 0) {
        return 1;
    }
    dim3 grid(N / L_NB), block(256);
    if (ver >= 2 && C <= 384) {
        static const int pf = getenv("MIBC_LSTM_PF") ? atoi(getenv("MIBC_LSTM_PF")) : 4;
        switch (C) {
            case 128:
                hipLaunchKernelGGL((lstm_layer_v2_kernel<128, 8>), grid, block, 0, s, Xin, Xout, Wf, biasn, T, N, reverse);
                return 0;
            case 256:
                hipLaunchKernelGGL((lstm_layer_v2_kernel<256, 8>), grid, block, 0, s, Xin, Xout, Wf, biasn, T, N, reverse);
                return 0;
            case 384:
                if (pf == 6)
                    hipLaunchKernelGGL((lstm_layer_v2_kernel<384, 6>), grid, block, 0, s, Xin, Xout, Wf, biasn, T, N, reverse);
                else if (pf == 3)
                    hipLaunchKernelGGL((lstm_layer_v2_kernel<384, 3>), grid, block, 0, s, Xin, Xout, Wf, biasn, T, N, reverse);
                else
                    hipLaunchKernelGGL((lstm_layer_v2_kernel<384, 4>), grid, block, 0, s, Xin, Xout, Wf, biasn, T, N, reverse);
                return 0;
            default:
                break;
        }
    }
#define LSTM_CASE(CC)                                                                          \
    case CC:                                                                                   \
        hipLaunchKernelGGL((lstm_layer_kernel<CC>), grid, block, 0, s, Xin, Xout, Wf, biasf, T, N, \
                           reverse);                                                           \
        return 0;
    switch (C) {
        LSTM_CASE(128)
        LSTM_CASE(256)
        LSTM_CASE(384)
        LSTM_CASE(512)
        default:
            return 1;
    }
#undef LSTM_CASE
}
