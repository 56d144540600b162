// dorado_amd/csrc/gemm.hip — f16 MFMA GEMM with row maps, used for
//   (a2) conv3 as an implicit-im2col GEMM: out[t][n][:] = tanh(W3 . a2p[n][stride*t ..][:] + b)
//        (replaces host_linear "cutlass_conv", dorado/nn/ConvStack.cpp:241-261), and
//   (a4) the linear CRF head: scores[n][t][:] = [5 tanh](Wl . x[t][n][:] (+ b))
//        (replaces host_linear at dorado/nn/CRFModules.cpp:112-117).
//
// C[m][c] = act( sum_k A(m)[k] * B[c][k] + bias[c] ),  A(m) = A + (m / a_div) * a_outer +
// (m % a_div) * a_inner (halfs), both operands K-contiguous ("TN").  The output row map has the
// same form.  Workgroup tile 128 x 128, BK = 32, 4 waves (2 x 2), each wave 64 x 64 as 2 x 2
// v_mfma_f32_32x32x16_f16 tiles; weights are the MFMA A operand so that every lane owns 4
// consecutive output columns of one row (packed 8-byte LDS writes), and the tile leaves through
// LDS as full 256-byte rows (16-byte stores per lane) — the head's scores are the largest HBM
// stream of the LSTM models (2 KB/step for hac).
#include "common.h"

#include <type_traits>
#include "engine.h"
#include <stdlib.h>

// struct GemmArgs: engine.h (ONE definition shared with the callers)

#define G_BM 128
#define G_BN 128
#define G_BK 32
#define G_LD 40    // LDS row stride (halfs) for the K tiles: 80 B
#define G_CLD 136  // LDS row stride (halfs) for the output tile: 272 B

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) half_t lds[G_BM * G_CLD];  // 34816 B; K tiles alias it
    half_t *As = lds;                 // [128][40]
    half_t *Bs = lds + G_BM * G_LD;   // [128][40]
    half_t *Cs = lds;                 // [128][136]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware 1-D grid: block b runs on XCD b % 8 (observed placement, speed only).  All column
    // tiles of one row tile are consecutive blocks of the SAME XCD so the activation tile is
    // fetched into one L2 once and re-used for every column tile.
    const int ncol = p.Ncols / G_BN;
    const int nrow = (p.M + G_BM - 1) / G_BM;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rowtile = (j / ncol) * 8 + xcd;
    if (rowtile >= nrow) {
        return;
    }
    const int m0 = rowtile * G_BM;
    const int c0 = (j % ncol) * G_BN;

    // staging assignment: row = tid >> 1, 16-half segment = tid & 1
    const int lrow = tid >> 1, lseg = tid & 1;
    int am = m0 + lrow;
    if (am >= p.M) am = p.M - 1;
    const half_t *a_src = p.A + (long)(am / p.a_div) * p.a_outer + (long)(am % p.a_div) * p.a_inner +
                          lseg * 16;
    const half_t *b_src = p.B + (long)(c0 + lrow) * p.K + lseg * 16;

    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    half8_t ra0 = *(const half8_t *)(a_src);
    half8_t ra1 = *(const half8_t *)(a_src + 8);
    half8_t rb0 = *(const half8_t *)(b_src);
    half8_t rb1 = *(const half8_t *)(b_src + 8);

    const int nk = p.K / G_BK;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous tile's fragment reads are done
        *(half8_t *)(As + lrow * G_LD + lseg * 16) = ra0;
        *(half8_t *)(As + lrow * G_LD + lseg * 16 + 8) = ra1;
        *(half8_t *)(Bs + lrow * G_LD + lseg * 16) = rb0;
        *(half8_t *)(Bs + lrow * G_LD + lseg * 16 + 8) = rb1;
        __syncthreads();
        if (kt + 1 < nk) {
            const int ko = (kt + 1) * G_BK;
            ra0 = *(const half8_t *)(a_src + ko);
            ra1 = *(const half8_t *)(a_src + ko + 8);
            rb0 = *(const half8_t *)(b_src + ko);
            rb1 = *(const half8_t *)(b_src + ko + 8);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = *(const half8_t *)(Bs + (wn * 64 + i * 32 + (lane & 31)) * G_LD + ks * 16 +
                                           8 * (lane >> 5));
                xf[i] = *(const half8_t *)(As + (wm * 64 + i * 32 + (lane & 31)) * G_LD + ks * 16 +
                                           8 * (lane >> 5));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32x32x16(wf[i], xf[j], acc[i][j]);
        }
    }
    __syncthreads();  // all K-tile reads done before Cs (aliasing) is written

    // epilogue: D rows = output columns (weights), D cols = output rows m
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mloc = wm * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cloc = wn * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
                half4_t h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][q * 4 + e];
                    if (p.bias != nullptr) v += p.bias[c0 + cloc + e];
                    if (p.act == 3) {
                        v = 5.0f * fast_tanh(v);
                    } else if (p.act >= 0) {
                        v = act_apply(v, p.act);
                    }
                    h[e] = (half_t)v;
                }
                *(half4_t *)(Cs + mloc * G_CLD + cloc) = h;
            }
        }
    }
    __syncthreads();
    if (p.epi_mode == 2) {
        // SwiGLU: 64 output features per tile
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = tid + 256 * pass;
            const int row = c >> 3, seg = c & 7;
            const int m = m0 + row;
            if (m < p.M) {
                const half8_t y = *(const half8_t *)(Cs + row * G_CLD + seg * 8);
                const half8_t g = *(const half8_t *)(Cs + row * G_CLD + 64 + seg * 8);
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gf = (float)g[e];
                    o[e] = (half_t)(gf * fast_sigmoid(gf) * (float)y[e]);
                }
                half_t *dst = p.out + (long)(m / p.o_div) * p.o_outer + (long)(m % p.o_div) * p.o_inner +
                              (c0 >> 1) + seg * 8;
                *(half8_t *)dst = o;
            }
        }
        return;
    }
    if (p.epi_mode == 1 && p.vT != nullptr && c0 >= p.rope_cols) {
        // V third of the QKV projection: store TRANSPOSED, vT[n][h][d][t] (t contiguous), so the
        // attention kernel can stage its P.V operand with plain coalesced 16-byte loads.
        // Requires rope_T % 128 == 0 (a 128-row tile never straddles two chunks).
        const int n = m0 / p.rope_T, t0 = m0 % p.rope_T;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int c = tid + 256 * pass;          // 128 columns x 16 groups of 8 tokens
            const int col = c >> 4, tg = c & 15;
            half8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Cs[(tg * 8 + e) * G_CLD + col];
            const int cv = c0 - p.rope_cols + col;   // column inside V: h*64 + d
            *(half8_t *)(p.vT + ((size_t)n * (p.Ncols - p.rope_cols) + cv) * p.rope_T + t0 + tg * 8) = v;
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int c = tid + 256 * pass;
        const int row = c >> 4, seg = c & 15;
        const int m = m0 + row;
        if (m < p.M && (p.ncols_valid == 0 || c0 + seg * 8 < p.ncols_valid)) {
            half_t *dst = p.out + (long)(m / p.o_div) * p.o_outer + (long)(m % p.o_div) * p.o_inner +
                          c0 + seg * 8;
            half8_t v = *(const half8_t *)(Cs + row * G_CLD + seg * 8);
            if (p.epi_mode == 1 && c0 + seg * 8 < p.rope_cols) {
                // rotary: partner 8 columns are 32 columns away inside the same 64-wide head
                const int cin = (seg * 8) & 63;               // column inside the head
                const bool lo = cin < 32;
                const half8_t w = *(const half8_t *)(Cs + row * G_CLD + seg * 8 + (lo ? 32 : -32));
                const float2 *tab = (const float2 *)p.rope + (size_t)(m % p.rope_T) * 32 + (cin & 31);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float2 cs = tab[e];
                    const float a = (float)v[e], b = (float)w[e];
                    // evens' = cos*e - sin*o ; odds' = sin*e + cos*o   (nn/TxModules.cpp:241-244); explicit fma forms
                    // (shared with gemm256.hip) so that both kernels round identically
                    v[e] = (half_t)(lo ? fmaf(cs.x, a, -(cs.y * b)) : fmaf(cs.y, b, cs.x * a));
                }
            }
            *(half8_t *)dst = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// gemm_dma_kernel: same tile / epilogue, but the operand tiles go HBM/L2 -> LDS by direct DMA
// (global_load_lds_dwordx4, 16 B per lane, no staging registers), BK = 64, two LDS stages so the
// DMA of tile k+1 overlaps the MFMAs of tile k.  The DMA writes LDS lane-linearly, so the 128-byte
// rows cannot be padded; bank conflicts are avoided instead by an XOR swizzle of the 16-byte column
// inside each row (phys = col ^ ((row >> 1) & 7)), applied on the SOURCE address when the tile is
// fetched and on the fragment read (cdna_hip_programming.md §5 "both sides or neither").
// ---------------------------------------------------------------------------------------------
#define D_BK 32
#define D_NST 4                      // LDS stages: 3 K tiles in flight behind the one being multiplied
#define D_TILE (G_BM * D_BK)         // halfs per operand tile (8 KiB)
#define D_STAGE (2 * D_TILE)         // A + B (16 KiB)

__device__ __forceinline__ void dma16(const half_t *g, half_t *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) half_t lds[D_NST * D_STAGE];  // 65536 B
    half_t *Cs = lds;                                                     // [128][136] aliases the stages

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = p.Ncols / G_BN;
    const int nrow = (p.M + G_BM - 1) / G_BM;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rowtile = (j / ncol) * 8 + xcd;
    if (rowtile >= nrow) {
        return;
    }
    const int m0 = rowtile * G_BM;
    const int c0 = (j % ncol) * G_BN;

    // DMA assignment (64-byte rows): instruction q of this wave fills 16-byte slots
    // [(wave*2+q)*64, +64) of the tile: row = (wave*2+q)*16 + lane/4, physical column lane%4 <-
    // logical column (lane%4) ^ ((row >> 2) & 3)
    const half_t *a_ptr[2], *b_ptr[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        const int col = (lane & 3) ^ ((row >> 2) & 3);
        int am = m0 + row;
        if (am >= p.M) am = p.M - 1;
        a_ptr[q] = p.A + (long)(am / p.a_div) * p.a_outer + (long)(am % p.a_div) * p.a_inner + col * 8;
        b_ptr[q] = p.B + (long)(c0 + row) * p.K + col * 8;
    }
    auto issue = [&](int kt) {
        half_t *As = lds + (kt % D_NST) * D_STAGE, *Bs = As + D_TILE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            dma16(a_ptr[q] + kt * D_BK, As + (wave * 2 + q) * 512);
            dma16(b_ptr[q] + kt * D_BK, Bs + (wave * 2 + q) * 512);
        }
    };

    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;

    const int nk = p.K / D_BK;
#pragma unroll
    for (int s0 = 0; s0 < D_NST - 1; ++s0)
        if (s0 < nk) issue(s0);
    int wrow[2], xrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        wrow[i] = wn * 64 + i * 32 + (lane & 31);
        xrow[i] = wm * 64 + i * 32 + (lane & 31);
    }
    const int lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed when at most the (up to 2) younger stages' DMAs (4 per wave each) are
        // still outstanding; a raw barrier (no vmcnt(0) drain) then publishes it to all waves and
        // also proves everybody is done reading the slot that stage kt+3 is about to overwrite
        if (kt + 2 < nk) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (kt + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + D_NST - 1 < nk && !(p.dbg & 4)) issue(kt + D_NST - 1);
        const half_t *As = lds + (kt % D_NST) * D_STAGE, *Bs = As + D_TILE;
#pragma unroll
        for (int ks = 0; ks < D_BK / 16; ++ks) {
            half8_t wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = *(const half8_t *)(Bs + wrow[i] * D_BK + (((2 * ks + lhi) ^ ((wrow[i] >> 2) & 3)) << 3));
                xf[i] = *(const half8_t *)(As + xrow[i] * D_BK + (((2 * ks + lhi) ^ ((xrow[i] >> 2) & 3)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    if (!(p.dbg & 2)) acc[i][jj] = mfma32x32x16(wf[i], xf[jj], acc[i][jj]);
        }
    }
    __syncthreads();  // all stage reads done before Cs (aliasing) is written
    if (p.dbg & 1) {
        if (acc[0][0][0] == 123.456f) p.out[0] = (half_t)1.0f;
        return;
    }

    // epilogue: D rows = output columns (weights), D cols = output rows m.
    // The lane's 32 bias values are fetched in ONE batch and the activation switch sits outside the element loops: a
    // load + branch per element costs one exposed L2 round trip each (hipcc waits vmcnt(0) per load here).
    if (p.bias != nullptr) {
        float4_t bv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bv[i][q] = *(const float4_t *)(p.bias + c0 + wn * 64 + i * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][jj][q * 4 + e] += bv[i][q][e];
    }
    auto activate = [&](auto act_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][jj][r] = (ACT == 3) ? 5.0f * fast_tanh(acc[i][jj][r]) : act_apply(acc[i][jj][r], ACT);
    };
    if (p.act == 3) activate(std::integral_constant<int, 3>{});
    else if (p.act == 0) activate(std::integral_constant<int, 0>{});
    else if (p.act == 1) activate(std::integral_constant<int, 1>{});
    else if (p.act == 2) activate(std::integral_constant<int, 2>{});
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int mloc = wm * 64 + jj * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cloc = wn * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
                half4_t h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)acc[i][jj][q * 4 + e];
                *(half4_t *)(Cs + mloc * G_CLD + cloc) = h;
            }
        }
    }
    __syncthreads();
    if (p.epi_mode == 2) {
        // SwiGLU: 64 output features per tile
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int c = tid + 256 * pass;
            const int row = c >> 3, seg = c & 7;
            const int m = m0 + row;
            if (m < p.M) {
                const half8_t y = *(const half8_t *)(Cs + row * G_CLD + seg * 8);
                const half8_t g = *(const half8_t *)(Cs + row * G_CLD + 64 + seg * 8);
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gf = (float)g[e];
                    o[e] = (half_t)(gf * fast_sigmoid(gf) * (float)y[e]);
                }
                half_t *dst = p.out + (long)(m / p.o_div) * p.o_outer + (long)(m % p.o_div) * p.o_inner +
                              (c0 >> 1) + seg * 8;
                *(half8_t *)dst = o;
            }
        }
        return;
    }
    if (p.epi_mode == 1 && p.vT != nullptr && c0 >= p.rope_cols) {
        // V third of the QKV projection: store TRANSPOSED, vT[n][h][d][t] (t contiguous), so the
        // attention kernel can stage its P.V operand with plain coalesced 16-byte loads.
        // Requires rope_T % 128 == 0 (a 128-row tile never straddles two chunks).
        const int n = m0 / p.rope_T, t0 = m0 % p.rope_T;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int c = tid + 256 * pass;          // 128 columns x 16 groups of 8 tokens
            const int col = c >> 4, tg = c & 15;
            half8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Cs[(tg * 8 + e) * G_CLD + col];
            const int cv = c0 - p.rope_cols + col;   // column inside V: h*64 + d
            *(half8_t *)(p.vT + ((size_t)n * (p.Ncols - p.rope_cols) + cv) * p.rope_T + t0 + tg * 8) = v;
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int c = tid + 256 * pass;
        const int row = c >> 4, seg = c & 15;
        const int m = m0 + row;
        if (m < p.M && (p.ncols_valid == 0 || c0 + seg * 8 < p.ncols_valid)) {
            half_t *dst = p.out + (long)(m / p.o_div) * p.o_outer + (long)(m % p.o_div) * p.o_inner +
                          c0 + seg * 8;
            half8_t v = *(const half8_t *)(Cs + row * G_CLD + seg * 8);
            if (p.epi_mode == 1 && c0 + seg * 8 < p.rope_cols) {
                // rotary: partner 8 columns are 32 columns away inside the same 64-wide head
                const int cin = (seg * 8) & 63;               // column inside the head
                const bool lo = cin < 32;
                const half8_t w = *(const half8_t *)(Cs + row * G_CLD + seg * 8 + (lo ? 32 : -32));
                const float2 *tab = (const float2 *)p.rope + (size_t)(m % p.rope_T) * 32 + (cin & 31);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float2 cs = tab[e];
                    const float a = (float)v[e], b = (float)w[e];
                    // evens' = cos*e - sin*o ; odds' = sin*e + cos*o   (nn/TxModules.cpp:241-244); explicit fma forms
                    // (shared with gemm256.hip) so that both kernels round identically
                    v[e] = (half_t)(lo ? fmaf(cs.x, a, -(cs.y * b)) : fmaf(cs.y, b, cs.x * a));
                }
            }
            *(half8_t *)dst = v;
        }
    }
}

extern "C" int mibc_launch_gemm256(hipStream_t s, const GemmArgs *a);
extern "C" int mibc_launch_gemm256x(hipStream_t s, const GemmArgs *a);

extern "C" int mibc_launch_gemm_tn(hipStream_t s, const GemmArgs *a) {
    if (a->K % G_BK != 0 || a->Ncols % G_BN != 0 || a->M <= 0) {
        return 1;
    }
    // large row counts with K = 512 / 1024 / 2048: the persistent 256 x 256 tile kernel (gemm256.hip; same
    // arithmetic, bit-identical results).  dbg bit 8 (microbenchmark) keeps the 128 x 128 kernel.
    // (round 4) gemm256x.hip first: the same tile on 16 x 16 x 32 MFMAs (higher sustained clock), plain / RoPE epilogues.
    // dbg (debug build, microbenchmarks): 0x1000 | bits = gemm256_kernel, 0x2000 | bits = gemm256x_kernel, 0x100 = 128 x 128.
    if ((a->dbg == 0 || (a->dbg & 0xf000) == 0x2000) && mibc_launch_gemm256x(s, a) == 0) return 0;
    if ((a->dbg == 0 || (a->dbg & 0xf000) == 0x1000) && mibc_launch_gemm256(s, a) == 0) return 0;
    const int ncol = a->Ncols / G_BN;
    const int nrow = (a->M + G_BM - 1) / G_BM;
    dim3 grid(((nrow + 7) / 8) * 8 * ncol);
    static const int use_dma = MIBC_ENV_INT("MIBC_GEMM_DMA", 1);
    if (use_dma && a->K % D_BK == 0) {
        hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, s, *a);
        return 0;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, s, *a);
    return 0;
}
