// dorado_amd/csrc/gemm256.hip — persistent 256 x 256 tile f16 MFMA GEMM for the large row counts of the hot path:
// the linear CRF head of the wide LSTM models (K = 1024) and every dense layer of the transformer models
// (QKV + RoPE, out-proj, FC1 + SwiGLU, FC2, upsample, CRF: K = 512 / 2048).  Same contract as gemm.hip
//      C[m][c] = epi( sum_k A(m)[k] * B[c][k] + bias[c] )       (both operands K-contiguous, row maps as GemmArgs)
// and the same arithmetic, element for element (v_mfma_f32_32x32x16_f16, weights as the A operand, k ascending),
// so results are bit-identical to gemm_dma_kernel; what changes is the machine mapping, taken from the cluster
// LSTM kernel (lstm_cluster.hip), where it was measured:
//   * one workgroup per CU (persistent, XCD-aware tile order), 8 waves = 4 (rows) x 2 (columns), wave tile
//     64 rows x 128 columns = 8 accumulator tiles of 32 x 32 (128 registers);
//   * both operands HBM/L2 -> LDS by direct DMA (global_load_lds_dwordx4) through a 4-slot ring of K = 32 slabs,
//     XOR-swizzled 64-byte rows, counted vmcnt, raw barriers; the slab stream runs across tile boundaries (the
//     first slabs of the next tile are in flight during the epilogue);
//   * the K loop of a tile is fully unrolled (ring slots and slab offsets are compile-time constants: eight waves
//     share the CU's scalar issue) and the two wave groups {0-3}, {4-7} run in anti-phase — one feeds the matrix
//     pipe while its SIMD partner occupies the LDS / DMA issue (group B executes one extra barrier up front);
//   * epilogues are lane-local on the accumulators: bias + activation; rotary embedding (a head's two halves are
//     the accumulator tiles g, g+1 of one lane) with the V third stored transposed for the attention kernel;
//     SwiGLU (y = tiles 0,1, gate = tiles 2,3); rows leave through per-wave LDS patches as 16-byte stores.
#include "common.h"
#include "engine.h"

#include <utility>

#define G2_BK 32
#define G2_NST 4
#define G2_TILE (256 * G2_BK)             // halfs per operand slab (16 KiB)
#define G2_STAGE (2 * G2_TILE)            // halfs per stage (32 KiB): weights | activations
#define G2_PATCH_LD 40
#define G2_PATCH (32 * G2_PATCH_LD)
#define G2_OFF_PATCH (G2_NST * G2_STAGE * 2)
#define G2_OFF_BIAS (G2_OFF_PATCH + 8 * G2_PATCH * 2)     // [2][256] f32: bias of the current / previous tile's columns
#define G2_LDS_BYTES (G2_OFF_BIAS + 2 * 256 * 4)

#define LDSP(T) __attribute__((address_space(3))) T *
typedef __attribute__((address_space(3))) void *g2_lds_vptr;
typedef const __attribute__((address_space(1))) half_t *g2_ghalf_p;

__device__ __forceinline__ void g2_dma16f(const float *g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (g2_lds_vptr)(size_t)lds_addr, 16, 0, 0);
}
__device__ __forceinline__ void g2_dma16(g2_ghalf_p g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (g2_lds_vptr)(size_t)lds_addr, 16, 0, 0);
}

template <typename F, int... I>
__device__ __forceinline__ void g2_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void g2_static_for(F &&f) {
    g2_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// EPI: 0 = bias / activation (GemmArgs::act), 1 = rotary embedding on columns < rope_cols + transposed V store,
//      2 = SwiGLU (64 y | 64 gate columns per 128: writes 64 columns).
// DBG (debug build only, wrong results): 1 no epilogue stores, 2 no epilogue at all, 4 A rows of every tile taken from
// tile 0 (cache-resident), 8 no MFMA, 16 single wave group epilogue timing probe (unused)
template <int KS, int EPI, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDSP(unsigned char) smem3 = (LDSP(unsigned char))smem;
    LDSP(half_t) stage = (LDSP(half_t))smem3;
    const unsigned lds0 = (unsigned)(size_t)smem3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool grpB = wave >= 4;
    LDSP(half_t) patch = (LDSP(half_t))(smem3 + G2_OFF_PATCH) + wave * G2_PATCH;

    // ---- persistent tile order: XCD x owns row tiles x, x + 8, ...; the workgroups of an XCD walk that list
    // column tile fastest, so the column tiles of a row tile run together on one L2 (observed placement: block b
    // runs on XCD b % 8; speed only) ----
    const int ncol = p.Ncols / 256;
    const int nrow = (p.M + 255) / 256;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int rows_x = (nrow - xcd + 7) / 8;              // row tiles of this XCD
    const int ntile_x = rows_x * ncol;
    auto tile_rc = [&](int i, int &rowtile, int &c0) __attribute__((always_inline)) {
        const int idx = slot + nslot * i;                 // index in this XCD's list
        rowtile = (idx / ncol) * 8 + xcd;
        c0 = (idx % ncol) * 256;
    };
    const int my_tiles = (ntile_x > slot) ? (ntile_x - slot + nslot - 1) / nslot : 0;
    if (my_tiles == 0) return;

    // DMA assignment (both operands): instruction q of this wave fills 16-byte slots [(wave*2+q)*64, +64) of a slab:
    // row = (wave*2+q)*16 + lane/4, physical 16-byte column lane%4 <- logical column (lane%4) ^ ((row>>2)&3)
    int drow[2], dcol[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        drow[q] = (wave * 2 + q) * 16 + (lane >> 2);
        dcol[q] = ((lane & 3) ^ ((drow[q] >> 2) & 3)) * 8;
    }
    const unsigned dma_lds = lds0 + (unsigned)(wave * 2) * 1024u;
    const int sw = (l31 >> 2) & 3;
    const int c0f = ((0 + lhi) ^ sw) << 3, c1f = ((2 + lhi) ^ sw) << 3;
    const int woff = (wn * 128 + l31) * G2_BK, xoff = G2_TILE + (wm * 64 + l31) * G2_BK;

    // per-tile DMA sources: uniform base pointers + this lane's constant byte offsets
    unsigned long long a_cur = 0, b_cur = 0, a_nxt = 0, b_nxt = 0;   // uniform (bases of A rows block / B cols block)
    unsigned aoffb[2] = {0, 0}, boffb[2], aoffb_n[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) boffb[q] = (unsigned)((drow[q] * p.K + dcol[q]) * 2);
    auto tile_src = [&](int rowtile, int c0, unsigned long long &ab, unsigned long long &bb, unsigned (&ao)[2]) __attribute__((always_inline)) {
        // A: rows m0 + drow (clamped), arbitrary row map -> per-lane offsets relative to the first row of the tile
        const int m0 = (DBG & 4) ? 0 : rowtile * 256;
        int mb = m0 < p.M ? m0 : p.M - 1;
        const long base0 = (long)(mb / p.a_div) * p.a_outer + (long)(mb % p.a_div) * p.a_inner;
        ab = (unsigned long long)(p.A + base0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int am = m0 + drow[q];
            if (am >= p.M) am = p.M - 1;
            const long off = (long)(am / p.a_div) * p.a_outer + (long)(am % p.a_div) * p.a_inner - base0;
            ao[q] = (unsigned)((off + dcol[q]) * 2);
        }
        bb = (unsigned long long)(p.B + (long)c0 * p.K);
    };
    auto issue = [&](int slot_, unsigned long long ab, unsigned long long bb, const unsigned (&ao)[2], int kslab) __attribute__((always_inline)) {
        const unsigned l = dma_lds + (unsigned)slot_ * (G2_STAGE * 2);
        ab += (unsigned)kslab * (G2_BK * 2);
        bb += (unsigned)kslab * (G2_BK * 2);
        asm volatile("" : "+s"(ab));
        asm volatile("" : "+s"(bb));
        g2_dma16((g2_ghalf_p)(bb + boffb[0]), l);
        g2_dma16((g2_ghalf_p)(bb + boffb[1]), l + 1024);
        g2_dma16((g2_ghalf_p)(ab + ao[0]), l + G2_TILE * 2);
        g2_dma16((g2_ghalf_p)(ab + ao[1]), l + G2_TILE * 2 + 1024);
    };

    int rowtile = 0, c0 = 0, rowtile_p = 0, c0_p = 0;
    tile_rc(0, rowtile, c0);
    tile_src(rowtile, c0, a_cur, b_cur, aoffb);
    issue(0, a_cur, b_cur, aoffb, 0);
    issue(1, a_cur, b_cur, aoffb, 1);

    float16_t acc[4][2];
    half8_t wf[4], xa[2];
    // Epilogue of tile (e_rowtile, e_c0).  D row = output column n = c0 + wn*128 + g*32 + (r&3) + 8 (r>>2) + 4 lhi ;
    // D col = output row m0 + wm*64 + rt*32 + l31.
    // e_zone: which half of the LDS bias zone holds this tile's 256 bias values (EPI 0 with a bias vector: fetched by
    // LDS-DMA during the tile's K loop — a VGPR-returning load here would make hipcc drain vmcnt(0), i.e. the whole
    // DMA pipeline, per element)
    auto epilogue = [&](int e_rowtile, int e_c0, int e_zone) __attribute__((always_inline)) {
        if (DBG & 2) {
            asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));
            return;
        }
        // D row = output column n = c0 + wn*128 + g*32 + (r&3) + 8 (r>>2) + 4 lhi ; D col = output row m0 + wm*64 + rt*32 + l31
        const int m0 = e_rowtile * 256;
        const int cw = e_c0 + wn * 128;                       // first column of this wave
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int mrow = m0 + wm * 64 + rt * 32;        // first of the 32 rows of this accumulator column block
            const int mmine = mrow + l31;                    // the row whose values this lane holds
            if (EPI == 1 && cw < p.rope_cols) {
                // rotary embedding: head = 64 columns = tiles (g, g+1); first half rotates with the second
                // (nn/TxModules.cpp:232-244: "evens/odds" = first/second half of head_dim)
                const float2 *tab = (const float2 *)p.rope + (size_t)(mmine % p.rope_T) * 32;
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int hr = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const float2 cs = tab[hr];
                        const float a = (float)(half_t)acc[2 * hp][rt][r], b = (float)(half_t)acc[2 * hp + 1][rt][r];
                        acc[2 * hp][rt][r] = fmaf(cs.x, a, -(cs.y * b));
                        acc[2 * hp + 1][rt][r] = fmaf(cs.y, a, cs.x * b);
                    }
                }
            }
            if (EPI == 1 && p.vT != nullptr && cw >= p.rope_cols) {
                // V third of the QKV projection: store TRANSPOSED, vT[n][h][d][t] (t contiguous)
                const int nchunk = m0 / p.rope_T, t0 = m0 % p.rope_T + wm * 64 + rt * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int hr = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        patch[hr * G2_PATCH_LD + l31] = (half_t)acc[g][rt][r];      // [d][token]
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int dd = (lane >> 2) + 16 * i, seg = lane & 3;
                        const half8_t v = *(LDSP(const half8_t))(patch + dd * G2_PATCH_LD + seg * 8);
                        const int cv = cw - p.rope_cols + g * 32 + dd;            // column inside V: h*64 + d
                        *(half8_t *)(p.vT + ((size_t)nchunk * (p.Ncols - p.rope_cols) + cv) * p.rope_T + t0 + seg * 8) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                continue;
            }
            // ---- bias / activation (EPI 0), SwiGLU (EPI 2), then rows leave through the wave's LDS patch.  A store
            // instruction covers 8 rows x 128 B (whole cache lines): pairs of 32-column accumulator tiles are transposed
            // 16 rows at a time (the CU's store path costs ~100 cycles per instruction and per 16 row segments, whatever
            // their width: with 16 rows x 64 B per instruction the stores were a third of a K = 512 tile) ----
            constexpr int NG = (EPI == 2) ? 2 : 4;
            if (EPI == 0) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (p.bias != nullptr) {
                        LDSP(const float) bz = (LDSP(const float))(smem3 + G2_OFF_BIAS) + e_zone * 256 + wn * 128 + g * 32 + 4 * lhi;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4_t bv = *(LDSP(const float4_t))(bz + 8 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[g][rt][q * 4 + e] += bv[e];
                        }
                    }
                    // the activation code is a wave-uniform switch OUTSIDE the element loops
                    if (p.act == 3) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[g][rt][r] = 5.0f * fast_tanh(acc[g][rt][r]);
                    } else if (p.act == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[g][rt][r] = act_apply(acc[g][rt][r], 0);
                    } else if (p.act == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[g][rt][r] = act_apply(acc[g][rt][r], 1);
                    } else if (p.act == 2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[g][rt][r] = act_apply(acc[g][rt][r], 2);
                    }
                }
            }
            // rows this lane stores after the transposition of half hh: mrow + 16 hh + lane/8 + 8 i
            half_t *orow2[2][2];
            bool ook2[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = mrow + 16 * hh + (lane >> 3) + 8 * i;
                    ook2[hh][i] = m < p.M;
                    const int mm = ook2[hh][i] ? m : 0;
                    orow2[hh][i] = p.out + (long)(mm / p.o_div) * p.o_outer + (long)(mm % p.o_div) * p.o_inner;
                }
            constexpr int NGP = NG / 2;
            constexpr int PLD = 72;   // halfs per patch row: 64 columns + 8 pad
#pragma unroll
            for (int gp = 0; gp < NGP; ++gp) {
                half4_t hv[2][4];
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int g = gp * 2 + g2;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = q * 4 + e;
                            float v = acc[g][rt][r];
                            if (EPI == 2) {
                                // SwiGLU (nn/TxModules.cpp:171-175): silu(gate) * y on the f16-rounded GEMM outputs
                                const float y = (float)(half_t)v, gt = (float)(half_t)acc[g + 2][rt][r];
                                v = gt * fast_sigmoid(gt) * y;
                            }
                            hv[g2][q][e] = (half_t)v;
                        }
                }
                const int ocol = (EPI == 2) ? ((e_c0 >> 1) + wn * 64) : (cw + gp * 64);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if ((l31 >> 4) == hh) {
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                *(LDSP(half4_t))(patch + (l31 & 15) * PLD + g2 * 32 + 8 * q + 4 * lhi) = hv[g2][q];
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int prow = (lane >> 3) + 8 * i, seg = lane & 7;
                        const half8_t v = *(LDSP(const half8_t))(patch + prow * PLD + seg * 8);
                        if (DBG & 1) {
                            asm volatile("" ::"v"(v));
                        } else if (ook2[hh][i]) {
                            *(half8_t *)(orow2[hh][i] + ocol + seg * 8) = v;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    };
    if (grpB) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma nounroll
    for (int ti = 0; ti < my_tiles; ++ti) {
        // next tile of this workgroup (for the two look-ahead slabs at the end of this tile's K loop); past the
        // last tile the current tile's first slabs are fetched again (nobody reads them)
        int rowtile_n = rowtile, c0_n = c0;
        if (ti + 1 < my_tiles) tile_rc(ti + 1, rowtile_n, c0_n);
        tile_src(rowtile_n, c0_n, a_nxt, b_nxt, aoffb_n);

        g2_static_for<KS>([&](auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int slot_ = ks & 3;
            // (KS % 4 == 0: slab g of the stream sits in ring slot ks & 3 in every tile)
            if (!grpB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // B1
            asm volatile("" ::: "memory");
            {
                LDSP(const half_t) sp = stage + slot_ * G2_STAGE;
#pragma unroll
                for (int g = 0; g < 4; ++g) wf[g] = *(LDSP(const half8_t))(sp + woff + g * 32 * G2_BK + c0f);
                xa[0] = *(LDSP(const half8_t))(sp + xoff + c0f);
                xa[1] = *(LDSP(const half8_t))(sp + xoff + 32 * G2_BK + c0f);
                constexpr int kt = ks + 2;
                if (kt < KS) issue(kt & 3, a_cur, b_cur, aoffb, kt);
                else issue(kt & 3, a_nxt, b_nxt, aoffb_n, kt - KS);
                // group A's epilogue of the PREVIOUS tile sits here, behind L(0) of this tile: it then runs beside
                // group B's M(KS-1) + epilogue of that tile (half a slab later by construction) instead of before it
                if (ks == 0 && !grpB && ti > 0) epilogue(rowtile_p, c0_p, (ti - 1) & 1);
                // this tile's bias values -> LDS zone ti & 1 (one extra DMA of wave 0: the counted waits only get stricter)
                if (EPI == 0 && ks == 1 && wave == 0 && p.bias != nullptr)
                    g2_dma16f(p.bias + c0 + lane * 4, lds0 + G2_OFF_BIAS + (unsigned)(ti & 1) * 1024u);
            }
            if (grpB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // B2
            asm volatile("" ::: "memory");
            {
                if (ks == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[g][rt][r] = 0.0f;
                }
                LDSP(const half_t) sp = stage + slot_ * G2_STAGE;
                half8_t xb[2];
                __builtin_amdgcn_s_setprio(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (!(DBG & 8)) {
                        acc[g][0] = mfma32x32x16(wf[g], xa[0], acc[g][0]);
                        acc[g][1] = mfma32x32x16(wf[g], xa[1], acc[g][1]);
                    } else {
                        asm volatile("" ::"v"(wf[g]), "v"(xa[0]), "v"(xa[1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    wf[g] = *(LDSP(const half8_t))(sp + woff + g * 32 * G2_BK + c1f);
                    if (g == 0) {
                        xb[0] = *(LDSP(const half8_t))(sp + xoff + c1f);
                        xb[1] = *(LDSP(const half8_t))(sp + xoff + 32 * G2_BK + c1f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (!(DBG & 8)) {
                        acc[g][0] = mfma32x32x16(wf[g], xb[0], acc[g][0]);
                        acc[g][1] = mfma32x32x16(wf[g], xb[1], acc[g][1]);
                    } else {
                        asm volatile("" ::"v"(wf[g]), "v"(xb[0]), "v"(xb[1]));
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                if (ks == KS - 1 && grpB) epilogue(rowtile, c0, ti & 1);
            }
        });

        rowtile_p = rowtile;
        c0_p = c0;
        rowtile = rowtile_n;
        c0 = c0_n;
        a_cur = a_nxt;
        b_cur = b_nxt;
        aoffb[0] = aoffb_n[0];
        aoffb[1] = aoffb_n[1];
    }
    if (!grpB) {
        epilogue(rowtile_p, c0_p, (my_tiles - 1) & 1);   // group A's last tile
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 0 = launched; 1 = shape not covered (caller uses gemm_dma_kernel).
extern "C" int mibc_launch_gemm256(hipStream_t s, const GemmArgs *a) {
#ifdef MIBC_DEBUG_KERNELS
    const int g2dbg = (a->dbg >= 0x1000) ? (a->dbg - 0x1000) : 0;
    if (a->Ncols % 256 != 0 || a->M < 2048 || a->ncols_valid != 0 || (a->dbg != 0 && a->dbg < 0x1000)) return 1;
#else
    if (a->Ncols % 256 != 0 || a->M < 2048 || a->ncols_valid != 0 || a->dbg != 0) return 1;
#endif
    if (a->K != 512 && a->K != 1024 && a->K != 2048) return 1;
    if (a->epi_mode == 1 && (a->rope_T % 256 != 0 || a->rope_cols % 128 != 0 || a->vT == nullptr)) return 1;
    const int ncu = mibc_ncu();   // of the launching thread's current device
    const long ntiles = (long)((a->M + 255) / 256) * (a->Ncols / 256);
    int grid = (ncu / 8) * 8;
    if (ntiles < grid) grid = (int)((ntiles + 7) / 8) * 8;
#define G2_LAUNCH(KS_, E_)                                                                                     \
    do {                                                                                                       \
        MIBC_LDS_ATTR_ONCE((gemm256_kernel<KS_, E_>), G2_LDS_BYTES);                                           \
        hipLaunchKernelGGL((gemm256_kernel<KS_, E_>), dim3(grid), dim3(512), G2_LDS_BYTES, s, *a);             \
        return 0;                                                                                              \
    } while (0)
#define G2_K(E_)                               \
    switch (a->K) {                            \
        case 512: G2_LAUNCH(16, E_);           \
        case 1024: G2_LAUNCH(32, E_);          \
        default: G2_LAUNCH(64, E_);            \
    }
#ifdef MIBC_DEBUG_KERNELS
#define G2_DBG(D_)                                                                                             \
    if (g2dbg == D_ && a->epi_mode == 0 && a->K == 512) {                                                     \
        (void)hipFuncSetAttribute((const void *)gemm256_kernel<16, 0, D_>,                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);                  \
        hipLaunchKernelGGL((gemm256_kernel<16, 0, D_>), dim3(grid), dim3(512), G2_LDS_BYTES, s, *a);          \
        return 0;                                                                                             \
    }
    G2_DBG(1) G2_DBG(2) G2_DBG(4) G2_DBG(6) G2_DBG(8) G2_DBG(10) G2_DBG(14)
#undef G2_DBG
#endif
    if (a->epi_mode == 1) { G2_K(1) }
    if (a->epi_mode == 2) { G2_K(2) }
    G2_K(0)
#undef G2_K
#undef G2_LAUNCH
}
