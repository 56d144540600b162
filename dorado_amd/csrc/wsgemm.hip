// dorado_amd/csrc/wsgemm.hip — "activation-stationary" MFMA GEMMs for the two big dense layers
// around the LSTM stack (SURVEY.md §8 a2 conv3, a4 CRF head).
//
// Both are [rows x K] . [K x cols] with a SMALL weight matrix (conv3: 384 x 304, head: 1024 x 384)
// and tens of millions of rows.  One workgroup keeps a 128-row activation tile resident in LDS for
// its whole life and streams the weights from L2 in MFMA-fragment order (fully coalesced 1 KiB
// wave loads, the same trick as lstm.hip), so every activation byte is read from HBM exactly once
// and never re-staged per column tile:
//   head : tile = X[t][n0 .. n0+128][0..C)  (contiguous 96 KiB for hac), out = scores[n][t][:]
//   conv3: tile = the contiguous a2p span of 128 consecutive output steps of one chunk
//          ((127*stride + W)*16 halfs = 25 KiB); im2col row t is the span at 96*t halfs.  The span
//          is stored in LDS with a 32-byte pad after every 192 bytes so that the 16 row-fragments
//          a ds_read_b128 group touches fall into 16 different bank groups (row stride 224 B).
// 8 waves (2 per SIMD), v_mfma_f32_16x16x32_f16, weights = A operand (16 output columns x 32 k),
// activations = B operand (16 rows x 32 k); each wave owns 64 output columns (4 column tiles) x
// all 128 rows per pass, i.e. every activation fragment read from LDS feeds 4 MFMAs.
// D layout: col = lane & 15 (row of the tile), row = 4*(lane >> 4) + r (output column) -> every
// lane stores 4 consecutive f16 outputs (8 bytes) of one row.
#include "common.h"

typedef float float4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4w mfma16(half8_t a, half8_t b, float4w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ half8_t wsload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

struct WsArgs {
    const half_t *A;     // activations
    const half_t *Wf;    // [cols/16][K/32][64][8] fragment order
    const float *bias;   // [cols] or nullptr
    half_t *out;
    int cols;            // multiple of 64
    int act;             // -1 none, 0/1/2 MIBC_ACT_*, 3 = 5*tanh
    // head: rows are (t, n'): tile index -> t = tile / tiles_per_t, n' = (tile % tiles_per_t)*128
    int N, Ns, n0, T;    // full batch, sub-batch, first chunk of the sub-batch, steps
    // conv3:
    int Tpitch, stride;  // a2p rows per chunk, conv stride
};

#define WS_ROWS 128

// MODE 0: head (A = X[T][N][C], K = C).  MODE 1: conv3 (A = a2p, K = 32*KT >= W*16).
template <int KT, int MODE>
__global__ __launch_bounds__(512, 2) void wsgemm_kernel(WsArgs p) {
    constexpr int K = KT * 32;
    constexpr int LD = K + 16;                      // MODE 0 row stride (halfs): conflict-free 16x32 reads
    constexpr int SPAN_BLK = 96;                    // MODE 1: halfs per 192-byte block
    extern __shared__ __attribute__((aligned(16))) half_t lds_all[];
    half_t *stage = lds_all;                 // 8 waves x 16 x 72 halfs = 18 KiB
    half_t *lds = lds_all + 8 * 16 * 72;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;

    int rows_valid = WS_ROWS;
    long out_base;     // MODE 0: n' of row 0 ; MODE 1: t of row 0
    int t_fix = 0, n_fix = 0;
    if (MODE == 0) {
        const int tiles_per_t = (p.Ns + WS_ROWS - 1) / WS_ROWS;
        t_fix = blockIdx.x / tiles_per_t;
        const int nloc = (blockIdx.x % tiles_per_t) * WS_ROWS;
        rows_valid = min(WS_ROWS, p.Ns - nloc);
        out_base = nloc;
        const half_t *src = p.A + ((size_t)t_fix * p.N + p.n0 + nloc) * K;
        // all global loads first, then the LDS stores (one memory latency, not one per iteration)
        constexpr int CH = WS_ROWS * (K / 8) / 512;
        half8_t stg_r[CH];
#pragma unroll
        for (int it = 0; it < CH; ++it) {
            const int c = tid + 512 * it;
            const int row = c / (K / 8), col8 = c % (K / 8);
            stg_r[it] = (half8_t)(0);
            if (row < rows_valid) stg_r[it] = *(const half8_t *)(src + (size_t)row * K + col8 * 8);
        }
#pragma unroll
        for (int it = 0; it < CH; ++it) {
            const int c = tid + 512 * it;
            *(half8_t *)(lds + (c / (K / 8)) * LD + (c % (K / 8)) * 8) = stg_r[it];
        }
    } else {
        const int tiles_per_chunk = (p.T + WS_ROWS - 1) / WS_ROWS;
        n_fix = blockIdx.x / tiles_per_chunk;
        const int t0 = (blockIdx.x % tiles_per_chunk) * WS_ROWS;
        rows_valid = min(WS_ROWS, p.T - t0);
        out_base = t0;
        // span of a2p rows [stride*t0, stride*t0 + 127*stride + K/16 + 1): 16 halfs per a2p row
        const half_t *src = p.A + ((size_t)n_fix * p.Tpitch + (size_t)p.stride * t0) * 16;
        const int span_halfs = (WS_ROWS - 1) * p.stride * 16 + K + 16;
        const long avail = ((long)p.Tpitch - (long)p.stride * t0) * 16;  // stay inside this chunk + slack
        constexpr int CH1 = 4;  // span <= 4 * 512 * 8 halfs
        half8_t stg_r[CH1];
        const int nch = (span_halfs + 7) / 8;
#pragma unroll
        for (int it = 0; it < CH1; ++it) {
            const int c = tid + 512 * it;
            stg_r[it] = (half8_t)(0);
            if (c < nch && (long)c * 8 + 8 <= avail) stg_r[it] = *(const half8_t *)(src + (size_t)c * 8);
        }
#pragma unroll
        for (int it = 0; it < CH1; ++it) {
            const int c = tid + 512 * it;
            const int h = c * 8;
            if (c < nch) *(half8_t *)(lds + (h / SPAN_BLK) * (SPAN_BLK + 16) + (h % SPAN_BLK)) = stg_r[it];
        }
    }
    __syncthreads();

    const int wvoff = lane * 16;
    const int passes = p.cols / 512 + ((p.cols % 512) ? 1 : 0);
    for (int pass = 0; pass < passes; ++pass) {
        const int col0 = pass * 512 + wave * 64;   // this wave's 64 columns
        if (col0 >= p.cols) break;
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(p.Wf + (size_t)(col0 / 16) * KT * 512), 0, 4 * KT * 1024, 0x00020000);
        float4w acc[4][8];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float4w bv = (float4w)(0.0f);
            if (p.bias != nullptr) bv = *(const float4w *)(p.bias + col0 + ct * 16 + 4 * lq);
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) acc[ct][rt] = bv;
        }
        half8_t wq[2][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wq[0][ct] = wsload(wrs, wvoff, (ct * KT + 0) * 1024);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            if (ks + 1 < KT) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) wq[(ks + 1) & 1][ct] = wsload(wrs, wvoff, (ct * KT + ks + 1) * 1024);
            }
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                half8_t b;
                if (MODE == 0) {
                    b = *(const half8_t *)(lds + (rt * 16 + l15) * LD + ks * 32 + 8 * lq);
                } else {
                    // row t = rt*16 + l15 starts at 96*t halfs of the span = block t (+ pad 8 halfs/block)
                    const int kb = ks * 32;                          // halfs into the row
                    const int blk = (rt * 16 + l15) * (p.stride * 16 / SPAN_BLK) + kb / SPAN_BLK;
                    b = *(const half8_t *)(lds + blk * (SPAN_BLK + 16) + (kb % SPAN_BLK) + 8 * lq);
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct][rt] = mfma16(wq[ks & 1][ct], b, acc[ct][rt]);
            }
        }
        // epilogue: lane holds out[row = rt*16 + l15][col0 + ct*16 + 4*lq .. +4].  Each wave
        // transposes 16 rows x 64 columns through its private 2 KiB LDS patch so that it leaves as
        // whole 128-byte row segments (8 rows per store instruction) instead of 32-byte pieces.
        half_t *stg = stage + wave * (16 * 72);   // [16 rows][64 + 8 pad]
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                half4_t h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[ct][rt][r];
                    if (p.act == 3) {
                        v = 5.0f * fast_tanh(v);
                    } else if (p.act >= 0) {
                        v = act_apply(v, p.act);
                    }
                    h[r] = (half_t)v;
                }
                *(half4_t *)(stg + l15 * 72 + ct * 16 + 4 * lq) = h;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r16 = half * 8 + (lane >> 3), seg = lane & 7;   // 8 rows x 8 x 16 B
                const int row = rt * 16 + r16;
                const half8_t v = *(const half8_t *)(stg + r16 * 72 + seg * 8);
                if (row < rows_valid) {
                    half_t *orow;
                    if (MODE == 0) {
                        orow = p.out + ((size_t)(out_base + row) * p.T + t_fix) * p.cols;
                    } else {
                        orow = p.out + ((size_t)(out_base + row) * p.N + n_fix) * p.cols;
                    }
                    *(half8_t *)(orow + col0 + seg * 8) = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// KT must be one of the instantiated values; returns 1 if the shape is not covered (caller falls
// back to gemm_tn_kernel).
extern "C" int mibc_launch_wsgemm(hipStream_t s, const WsArgs *a, int K, int mode) {
    if (a->cols % 64 != 0 || K % 32 != 0) return 1;
    const int KT = K / 32;
    size_t smem;
    int grid;
    if (mode == 0) {
        smem = (size_t)WS_ROWS * (K + 16) * 2 + 8 * 16 * 72 * 2;
        grid = a->T * ((a->Ns + WS_ROWS - 1) / WS_ROWS);
    } else {
        if ((a->stride * 16) % 96 != 0) return 1;  // pad scheme assumes 192-byte row pitch multiples
        const int span_halfs = (WS_ROWS - 1) * a->stride * 16 + K + 16;
        smem = (size_t)((span_halfs + 95) / 96 + 1) * 112 * 2 + 8 * 16 * 72 * 2;
        grid = a->N * ((a->T + WS_ROWS - 1) / WS_ROWS);
    }
    if (smem > 160 * 1024) return 1;
#define WS_CASE(KT_, M_)                                                                               \
    if (KT == KT_ && mode == M_) {                                                                     \
        static bool once = false;                                                                      \
        if (!once) {                                                                                   \
            (void)hipFuncSetAttribute((const void *)wsgemm_kernel<KT_, M_>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
            once = true;                                                                               \
        }                                                                                              \
        hipLaunchKernelGGL((wsgemm_kernel<KT_, M_>), dim3(grid), dim3(512), smem, s, *a);              \
        return 0;                                                                                      \
    }
    WS_CASE(4, 0) WS_CASE(8, 0) WS_CASE(12, 0) WS_CASE(16, 0)
    WS_CASE(10, 1)
#undef WS_CASE
    return 1;
}
