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
#include "engine.h"

typedef float float4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4w mfma16(half8_t a, half8_t b, float4w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ half8_t wsload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// Activation of four outputs.  All forms are  a + b / (1 + 2^(k x))  (one v_exp_f32, one
// v_rcp_f32 per element, the rest on the packed-f32 pipe):
//   3: 5 tanh x = 5 - 10 / (1 + 2^(2 log2e x))      2: tanh x = 1 - 2 / (1 + 2^(2 log2e x))
//   0: swish x = x / (1 + 2^(-log2e x))              1: min(swish x, 3.5)
//   4: tanh x, stored as int8 round(127 f16(tanh x)) — conv3 in front of the int8 LSTM (nn/ConvStack.cpp:243,324-329: the
//      reference converts in a separate host_convert pass; here the epilogue emits the int8 row, bit-identical to
//      q8_convert_kernel applied to the f16 output of act 2)
template <int act_>
__device__ __forceinline__ float4w ws_activation(float4w v) {
    constexpr int act = (act_ == 4) ? 2 : act_;
    if (act < 0) return v;
    const float k = (act >= 2) ? 2.88539008f : -1.44269504f;
    const float4w kv = v * (float4w)(k);
    float4w d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = __builtin_amdgcn_exp2f(kv[r]);
    d = d + (float4w)(1.0f);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = __builtin_amdgcn_rcpf(d[r]);
    // explicit fused multiply-adds: with -ffp-contract=fast the compiler is free to contract `d * a + b` in one instantiation and
    // not in another, and one f32 ulp flips an f16 rounding once per ~10^6 outputs — the int8 epilogue (act 4) must reproduce the
    // f16 epilogue (act 2) bit for bit (tests/test_gpu_lstm_q8.py)
    if (act == 3) return __builtin_elementwise_fma(d, (float4w)(-10.0f), (float4w)(5.0f));
    if (act == 2) return __builtin_elementwise_fma(d, (float4w)(-2.0f), (float4w)(1.0f));
    float4w sw = v * d;
    if (act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sw[r] = fminf(sw[r], 3.5f);
    }
    return sw;
}

// struct WsArgs: engine.h (ONE definition shared with the caller)


// MODE 0: head (A = X[T][N][C], K = C).  MODE 1: conv3 (A = a2p, K = 32*KT >= W*16).
// CT = 16-column tiles per wave per pass (8 waves x CT x 16 columns per pass).
//
// Persistent: the grid is one workgroup per CU (8 waves x 256 registers); each loops over tiles
// tile = blockIdx.x, += gridDim.x.  While tile i is being multiplied out of LDS, tile i+1 is already
// in flight from HBM into registers (pre[]), so the HBM latency of the activation read hides behind
// the MFMAs instead of being paid once per workgroup; the weight fragments of the next pass's first
// k-step are requested before the epilogue for the same reason.
template <int KT, int MODE, int CT, int RT, int NW, int ACT>
__global__ __launch_bounds__(NW * 64, 2) void wsgemm_kernel(WsArgs p) {
    constexpr int NT = NW * 64;
    constexpr int WS_ROWS = RT * 16;
    constexpr int K = KT * 32;
    constexpr int LD = K + 16;                      // MODE 0 row stride (halfs): conflict-free 16x32 reads
    constexpr int SPAN_BLK = 96;                    // MODE 1: halfs per 192-byte block
    constexpr int WCOLS = CT * 16;
    constexpr int CH = (MODE == 0) ? ((WS_ROWS * (K / 8) + NT - 1) / NT) : (2048 / NT);   // 16-byte chunks per thread per tile
    static_assert((KT & 1) == 0, "the weight double buffer assumes an even number of k-steps");
    extern __shared__ __attribute__((aligned(16))) half_t lds_all[];
    half_t *stage = lds_all;                 // NW waves x 16 x 72 halfs
    half_t *lds = lds_all + NW * 16 * 72;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;

    const int tiles_per_grp = (MODE == 0) ? ((p.Ns + WS_ROWS - 1) / WS_ROWS) : ((p.T + WS_ROWS - 1) / WS_ROWS);
    const int ntiles = (MODE == 0) ? (p.T * tiles_per_grp) : (p.N * tiles_per_grp);
    // MODE 1: span of a2p rows [stride*t0, stride*t0 + 127*stride + K/16 + 1): 16 halfs per a2p row
    const int span_halfs = (WS_ROWS - 1) * p.stride * 16 + K + 16;
    const int nch = (span_halfs + 7) / 8;

    half8_t pre[CH];
    auto tile_fetch = [&](int tile) {
        const int grp = tile / tiles_per_grp;
        const int r0 = (tile % tiles_per_grp) * WS_ROWS;
        if (MODE == 0) {
            const int rows_valid = min(WS_ROWS, p.Ns - r0);
            const half_t *src = p.A + ((size_t)grp * p.N + p.n0 + r0) * K;   // 128 consecutive rows
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const int c = tid + NT * it;
                pre[it] = (half8_t)(0);
                if (c < WS_ROWS * (K / 8) && c / (K / 8) < rows_valid)
                    pre[it] = (p.dbg & 8) ? __builtin_nontemporal_load((const half8_t *)(src + (size_t)c * 8)) : *(const half8_t *)(src + (size_t)c * 8);
            }
        } else {
            const half_t *src = p.A + ((size_t)grp * p.Tpitch + (size_t)p.stride * r0) * 16;
            const long avail = ((long)p.Tpitch - (long)p.stride * r0) * 16;  // stay inside this chunk + slack
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const int c = tid + NT * it;
                pre[it] = (half8_t)(0);
                if (c < nch && (long)c * 8 + 8 <= avail)
                    pre[it] = (p.dbg & 8) ? __builtin_nontemporal_load((const half8_t *)(src + (size_t)c * 8)) : *(const half8_t *)(src + (size_t)c * 8);
            }
        }
    };
    auto tile_commit = [&]() {
#pragma unroll
        for (int it = 0; it < CH; ++it) {
            const int c = tid + NT * it;
            if (MODE == 0) {
                if (c < WS_ROWS * (K / 8)) *(half8_t *)(lds + (c / (K / 8)) * LD + (c % (K / 8)) * 8) = pre[it];
            } else {
                const int h = c * 8;
                if (c < nch) *(half8_t *)(lds + (h / SPAN_BLK) * (SPAN_BLK + 16) + (h % SPAN_BLK)) = pre[it];
            }
        }
    };

    const int wvoff = lane * 16;
    const int passes = (p.cols + NW * WCOLS - 1) / (NW * WCOLS);
    auto wrsrc = [&](int pass) {
        const int col0 = pass * NW * WCOLS + wave * WCOLS;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.Wf + (size_t)(col0 / 16) * KT * 512), 0,
                                                 CT * KT * 1024, 0x00020000);
    };
    half8_t wq[2][CT];
    if (wave * WCOLS < p.cols) {
        const __amdgpu_buffer_rsrc_t w0 = wrsrc(0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wq[0][ct] = wsload(w0, wvoff, (ct * KT + 0) * 1024);
    }

    int tile = blockIdx.x;
    if (tile < ntiles) tile_fetch(tile);
    tile_commit();
    __syncthreads();

    for (; tile < ntiles; tile += gridDim.x) {
        const int grp = tile / tiles_per_grp;               // MODE 0: t ; MODE 1: chunk n
        const int out_base = (tile % tiles_per_grp) * WS_ROWS;   // MODE 0: n' of row 0 ; MODE 1: t of row 0
        const int rows_valid = min(WS_ROWS, ((MODE == 0) ? p.Ns : p.T) - out_base);
        const int tnext = tile + gridDim.x;
        if (tnext < ntiles) tile_fetch(tnext);

        const int dbg = p.dbg;
        for (int pass = 0; pass < passes; ++pass) {
            const int col0 = pass * NW * WCOLS + wave * WCOLS;   // this wave's columns
            if (col0 >= p.cols) break;
            const __amdgpu_buffer_rsrc_t wrs = wrsrc(pass);
            float4w acc[CT][RT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float4w bv = (float4w)(0.0f);
                if (p.bias != nullptr) bv = *(const float4w *)(p.bias + col0 + ct * 16 + 4 * lq);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[ct][rt] = bv;
            }
#pragma nounroll
            for (int ks0 = 0; ks0 < ((dbg & 4) ? 0 : KT); ks0 += 2) {
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const int ks = ks0 + uu;
                    const int kn = (ks + 1 < KT) ? ks + 1 : KT - 1;   // last one: harmless re-load
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        wq[(uu + 1) & 1][ct] = wsload(wrs, wvoff, (ct * KT + kn) * 1024);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
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
                        for (int ct = 0; ct < CT; ++ct) acc[ct][rt] = mfma16(wq[uu & 1][ct], b, acc[ct][rt]);
                    }
                }
            }
            {   // first weight fragments of the pass that runs next (this tile's or the next tile's)
                const int pn = (pass + 1 < passes && (pass + 1) * NW * WCOLS + wave * WCOLS < p.cols) ? pass + 1 : 0;
                const __amdgpu_buffer_rsrc_t wn = wrsrc(pn);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wq[0][ct] = wsload(wn, wvoff, (ct * KT + 0) * 1024);
            }
            // epilogue: lane holds out[row = rt*16 + l15][col0 + ct*16 + 4*lq .. +4].  Each wave
            // transposes 16 rows x WCOLS columns through its private LDS patch so that it leaves as
            // whole row segments (16 B per lane) instead of 8-byte pieces.  LDS operations of one
            // wave execute in order, so the only wait needed is write -> read.
            half_t *stg = stage + wave * (16 * 72);   // [16 rows][64 + 8 pad]
            if (ACT == 4) {
                // int8 rows: the patch holds 16 rows x WCOLS bytes (row pitch 80 B), a row leaves as WCOLS / 16 segments of 16 B
                unsigned char *stg8 = (unsigned char *)stg;
                constexpr int SEGS8 = WCOLS / 16;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const float4w h4 = ws_activation<ACT>(acc[ct][rt]);
                        int pk = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float f = fminf(1.0f, fmaxf(-1.0f, (float)(half_t)h4[r]));
                            pk |= ((int)__builtin_rintf(f * 127.0f) & 0xff) << (8 * r);
                        }
                        *(int *)(stg8 + l15 * 80 + ct * 16 + 4 * lq) = pk;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    {
                        const int r16 = lane / SEGS8, seg = lane % SEGS8;
                        const int row = rt * 16 + r16;
                        if (lane < 16 * SEGS8) {
                            const int4 v = *(const int4 *)(stg8 + r16 * 80 + seg * 16);
                            if (row < rows_valid && !(dbg & 2)) {
                                signed char *orow = (signed char *)p.out + ((size_t)(out_base + row) * p.N + grp) * p.cols;
                                *(int4 *)(orow + col0 + seg * 16) = v;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                continue;
            }
            constexpr int SEGS = WCOLS / 8;            // 16-byte segments per row
            constexpr int RPI = 64 / SEGS;             // rows per store instruction
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float4w h4 = (dbg & 1) ? acc[ct][rt] : ws_activation<ACT>(acc[ct][rt]);
                    half4_t h;
                    h[0] = (half_t)h4[0];
                    h[1] = (half_t)h4[1];
                    h[2] = (half_t)h4[2];
                    h[3] = (half_t)h4[3];
                    *(half4_t *)(stg + l15 * 72 + ct * 16 + 4 * lq) = h;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r16_0 = 0; r16_0 < 16; r16_0 += RPI) {
                    const int r16 = r16_0 + lane / SEGS, seg = lane % SEGS;
                    const int row = rt * 16 + r16;
                    if (lane < RPI * SEGS && r16 < 16) {
                        const half8_t v = *(const half8_t *)(stg + r16 * 72 + seg * 8);
                        if (row < rows_valid && !(dbg & 2)) {
                            half_t *orow;
                            if (MODE == 0) {
                                orow = p.out + ((size_t)(out_base + row) * p.T + grp) * p.cols;
                            } else {
                                orow = p.out + ((size_t)(out_base + row) * p.N + grp) * p.cols;
                            }
                            // head: streamed (nt) stores — the 56 GB of scores are not read back by this kernel (25.1 -> 24.2 ms on
                            // the hac batch, same box); conv3 keeps the default policy (nt measured 16.4 -> 17.5 ms)
                            if (MODE == 0) __builtin_nontemporal_store(v, (half8_t *)(orow + col0 + seg * 8));
                            else *(half8_t *)(orow + col0 + seg * 8) = v;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();   // every wave is done with this tile's activations
        if (tnext < ntiles) tile_commit();
        __syncthreads();
    }
}

// KT must be one of the instantiated values; returns 1 if the shape is not covered (caller falls
// back to gemm_tn_kernel).
extern "C" int mibc_launch_wsgemm(hipStream_t s, const WsArgs *a, int K, int mode) {
    if (a->cols % 64 != 0 || K % 32 != 0) return 1;
    const int KT = K / 32;
    // NW = waves per workgroup: 8 (one workgroup per CU) or 4 (two independent workgroups per CU,
    // so that one's MFMA k-loop runs beside the other's VALU epilogue)
    static const int ws_dbg = MIBC_ENV_INT("MIBC_WS_DBG", 0);
    WsArgs adbg = *a;
    if (ws_dbg) {
        adbg.dbg = ws_dbg;
        a = &adbg;
    }
    static const int nw_env = MIBC_ENV_INT("MIBC_WS_NW", 8);
    const int NW = (nw_env == 4) ? 4 : 8;
    // columns per wave per pass: whichever of 64 / 48 / 32 keeps all waves busy
    const int per = NW * 16;
    const int CT = (a->cols % (4 * per) == 0) ? 4 : (a->cols % (3 * per) == 0) ? 3 : (a->cols % (2 * per) == 0) ? 2 : 4;
    // rows per tile: 128, or fewer where the accumulators + the prefetched next tile do not fit the
    // 256-register budget / two workgroups do not fit the LDS
    int RT = 8;
    if (mode == 1 && CT == 4 && NW == 8) RT = 6;   // 128 rows x 64 columns per wave would spill
    if (mode == 0) {
        if (NW == 8) RT = (CT == 4) ? (KT >= 16 ? 5 : KT >= 8 ? 6 : 8) : 8;
        else RT = (KT >= 12) ? 4 : 8;
    }
    const int rows = RT * 16;
    size_t smem;
    int ntiles;
    if (mode == 0) {
        smem = (size_t)rows * (K + 16) * 2 + NW * 16 * 72 * 2;
        ntiles = a->T * ((a->Ns + rows - 1) / rows);
    } else {
        if ((a->stride * 16) % 96 != 0) return 1;  // pad scheme assumes 192-byte row pitch multiples
        const int span_halfs = (rows - 1) * a->stride * 16 + K + 16;
        if ((span_halfs + 7) / 8 > 2048) return 1;
        smem = (size_t)((span_halfs + 95) / 96 + 1) * 112 * 2 + NW * 16 * 72 * 2;
        ntiles = a->N * ((a->T + rows - 1) / rows);
    }
    if (smem > (size_t)(NW == 8 ? 160 : 80) * 1024) return 1;
    const int ncu = mibc_ncu();   // of the launching thread's current device
    const int slots = ncu * (NW == 8 ? 1 : 2);
    const int grid = ntiles < slots ? ntiles : slots;
    // activation is a template parameter (a run-time switch costs ~4 extra VALU per output): the
    // head uses 5*tanh (3) or none (-1), conv3 swish (0), clamped swish (1), tanh (2) or tanh with int8 rows (4)
#define WS_ONE(KT_, M_, CT_, RT_, NW_, ACT_)                                                           \
    if (a->act == ACT_) {                                                                              \
        MIBC_LDS_ATTR_ONCE((wsgemm_kernel<KT_, M_, CT_, RT_, NW_, ACT_>), 160 * 1024);                 \
        hipLaunchKernelGGL((wsgemm_kernel<KT_, M_, CT_, RT_, NW_, ACT_>), dim3(grid), dim3(NW_ * 64),  \
                           smem, s, *a);                                                               \
        return 0;                                                                                      \
    }
#define WS_CASE(KT_, M_, CT_, RT_, NW_)                                                                \
    if (KT == KT_ && mode == M_ && CT == CT_ && RT == RT_ && NW == NW_) {                              \
        if (M_ == 0) {                                                                                 \
            WS_ONE(KT_, M_, CT_, RT_, NW_, 3) WS_ONE(KT_, M_, CT_, RT_, NW_, -1)                       \
        } else {                                                                                       \
            WS_ONE(KT_, M_, CT_, RT_, NW_, 0) WS_ONE(KT_, M_, CT_, RT_, NW_, 1)                        \
            WS_ONE(KT_, M_, CT_, RT_, NW_, 2) WS_ONE(KT_, M_, CT_, RT_, NW_, 4)                        \
        }                                                                                              \
        return 1;                                                                                      \
    }
    WS_CASE(4, 0, 4, 8, 8) WS_CASE(8, 0, 4, 6, 8) WS_CASE(12, 0, 4, 6, 8) WS_CASE(16, 0, 4, 5, 8)
    WS_CASE(4, 0, 2, 8, 8) WS_CASE(8, 0, 2, 8, 8)
    WS_CASE(10, 1, 4, 6, 8) WS_CASE(10, 1, 3, 8, 8) WS_CASE(10, 1, 2, 8, 8)
    WS_CASE(10, 1, 3, 8, 4) WS_CASE(12, 0, 4, 4, 4)
#undef WS_CASE
#undef WS_ONE
    return 1;
}
