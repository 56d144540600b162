// dorado_amd/csrc/gemm256x.hip — round 4: the persistent 256 x 256 tile GEMM of gemm256.hip on v_mfma_f32_16x16x32_f16.
//      C[m][c] = epi( sum_k A(m)[k] * B[c][k] + bias[c] )       (both operands K-contiguous, row maps as GemmArgs)
// Why a second kernel: on random data the matrix pipe is POWER limited, and the 32 x 32 x 16 instruction draws more per
// flop than the 16 x 16 x 32 one — tools/mfma_tile_clock.hip (profiles/r04_mfma_tile_clock.jsonl), the per-slab mix of
// this very tile (12 fragment reads + the MFMAs of a 64 x 128 wave tile), 8 waves per CU, 100 ms runs: 1403 TFLOP/s at
// 1.57 GHz with 32x32x16, 1538 TFLOP/s at 1.91 GHz with 16x16x32; bare MFMAs 1635 vs 1863.  Same operand bytes per flop,
// same 128 accumulator registers.  What changes against gemm256.hip:
//   * 32 MFMAs per K = 32 slab and wave (8 weight tiles x 4 activation tiles of 16), weight fragments through a 3-deep
//     register ring read behind the MFMAs that free them;
//   * fragment rows are 16-row tiles: the LDS image keeps 64-byte rows with the 16-byte column XOR-ed by
//     (-(row >> 2)) & 3, which makes every ds_read_b128 lane group of the 16-row fragment pattern conflict-free
//     (MI355X_MICROARCH.md: groups {0-3, 12-15, 20-27}, ...: four row quads x four columns);
//   * D layout: a lane holds 4 CONSECUTIVE output columns of one row (col n = 16 g + 4 (lane >> 4) + r, row m = lane & 15):
//     rows leave through per-wave 8 x 128 LDS patches as 4 rows x 256 B per store instruction;
//   * tile order with COLUMN GROUPS: when the weight matrix does not fit an XCD's L2 (sup@v4.3 head: 4096 x 1024 = 8 MB),
//     an XCD owns `cg` column tiles (<= 2 MB of weights, L2 resident) and walks row tiles, instead of streaming all of B
//     from the Infinity Cache for every two row tiles (213 GB of L2 fills per head launch).
// Everything else (persistent workgroups, 4-slot LDS-DMA ring of K = 32 slabs, counted vmcnt, raw barriers, two wave
// groups in anti-phase, fully unrolled K loop, epilogue of group A behind the first DMA of the next tile) is gemm256.hip's,
// where it was measured.  Epilogues: 0 = bias / activation, 1 = rotary embedding + transposed V (QKV projection).
// Arithmetic: k ascending in steps of 32 per accumulator — NOT bit-identical to the 32 x 32 x 16 kernels (a different
// summation tree inside the instruction); tests/test_gpu_gemm256.py holds it to an f64 host product instead.
#include "common.h"
#include "engine.h"

#include <utility>

#define GX_BK 32
#define GX_NST 4
#define GX_TILE (256 * GX_BK)             // halfs per operand slab (16 KiB)
#define GX_STAGE (2 * GX_TILE)            // halfs per stage (32 KiB): weights | activations
#define GX_PLD 136                        // halfs per patch row: 128 + 8 pad (rows stay 16-byte aligned)
#define GX_VLD 72                         // halfs per patch row of the transposed-V gather (16 d rows x 64 tokens + 8 pad)
#define GX_PATCH (16 * GX_VLD)            // halfs per wave patch: max(8 x 136, 16 x 72) = 1152
#define GX_OFF_PATCH (GX_NST * GX_STAGE * 2)
#define GX_OFF_BIAS (GX_OFF_PATCH + 8 * GX_PATCH * 2)     // [2][256] f32: bias of the current / previous tile's columns
#define GX_LDS_BYTES (GX_OFF_BIAS + 2 * 256 * 4)

#define LDSP(T) __attribute__((address_space(3))) T *
typedef __attribute__((address_space(3))) void *gx_lds_vptr;
typedef const __attribute__((address_space(1))) half_t *gx_ghalf_p;

__device__ __forceinline__ void gx_dma16f(const float *g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (gx_lds_vptr)(size_t)lds_addr, 16, 0, 0);
}
__device__ __forceinline__ void gx_dma16(gx_ghalf_p g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (gx_lds_vptr)(size_t)lds_addr, 16, 0, 0);
}
__device__ __forceinline__ float4_t gx_mfma(half8_t a, half8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <typename F, int... I>
__device__ __forceinline__ void gx_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void gx_static_for(F &&f) {
    gx_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// EPI: 0 = bias / activation (GemmArgs::act), 1 = rotary embedding on columns < rope_cols + transposed V store.
// DBG (debug build only; results wrong except for 4): 1 no epilogue stores, 2 no epilogue at all, 4 output rows stored with
// the default cache policy instead of nt, 8 no MFMA, 16 / 32 weight / activation
// DMA pieces read contiguous 1 KB (what a pre-tiled slab image would give: whole 128-B lines instead of 16 half lines).
template <int KS, int EPI, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256x_kernel(GemmArgs p, int cg, int stagger) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDSP(unsigned char) smem3 = (LDSP(unsigned char))smem;
    LDSP(half_t) stage = (LDSP(half_t))smem3;
    const unsigned lds0 = (unsigned)(size_t)smem3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const bool grpB = wave >= 4;
    LDSP(half_t) patch = (LDSP(half_t))(smem3 + GX_OFF_PATCH) + wave * GX_PATCH;

    // ---- persistent tile order.  The column tiles form ncol / cg groups of cg tiles; a group belongs to xpg = 8 / groups
    // XCDs, which share its row tiles round-robin; the workgroups of an XCD walk their list column tile fastest.  cg = ncol
    // (one group) is gemm256.hip's order: XCD x owns row tiles x, x + 8, ... with all their columns.  (Observed placement:
    // block b runs on XCD b % 8; speed only.) ----
    const int ncol = p.Ncols / 256;
    const int nrow = (p.M + 255) / 256;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int ngrp = ncol / cg, xpg = 8 / ngrp;
    const int grp = xcd / xpg, xin = xcd % xpg;
    const int rows_x = (nrow - xin + xpg - 1) / xpg;     // row tiles of this XCD
    const int ntile_x = rows_x * cg;
    auto tile_rc = [&](int i, int &rowtile, int &c0) __attribute__((always_inline)) {
        const int idx = slot + nslot * i;                 // index in this XCD's list
        rowtile = (idx / cg) * xpg + xin;
        c0 = (grp * cg + idx % cg) * 256;
    };
    const int my_tiles = (ntile_x > slot) ? (ntile_x - slot + nslot - 1) / nslot : 0;
    if (my_tiles == 0) return;
    // Phase stagger: all workgroups run tiles of equal length, so without it every CU reaches its epilogue at the same
    // moment and the chip alternates between "nobody stores" and "256 CUs store 128 KB each" (the store burst then runs at
    // the HBM write rate and the waves sit in the next counted vmcnt wait behind their own stores).  Workgroup slot s starts
    // (s % 4) * stagger cycles late, so that a quarter of an XCD's CUs is in its epilogue at a time.
    if (stagger > 0) {
        const long long t0 = (long long)__builtin_readcyclecounter();
        const int smode = stagger & 3;          // experiment: which index sets the phase
        const long long wait = (long long)(smode == 0 ? ((slot / cg) & 3) : smode == 1 ? (xcd & 3) : smode == 2 ? ((slot / cg) & 7) : (((slot / cg) & 1) * 4)) * (long long)(stagger & ~3);
        while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }

    // DMA assignment (both operands): instruction q of this wave fills 16-byte slots [(wave*2+q)*64, +64) of a slab:
    // row = (wave*2+q)*16 + lane/4, physical 16-byte column lane%4 <- logical column (lane%4) ^ ((-(row>>2)) & 3)
    int drow[2], dcol[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        drow[q] = (wave * 2 + q) * 16 + (lane >> 2);
        dcol[q] = ((lane & 3) ^ ((0 - (drow[q] >> 2)) & 3)) * 8;
    }
    const unsigned dma_lds = lds0 + (unsigned)(wave * 2) * 1024u;
    // fragment read offsets (halfs): row l15 of a 16-row tile, logical 16-byte column lq
    const int pc8 = (lq ^ ((0 - (l15 >> 2)) & 3)) << 3;
    const int woff = (wn * 128 + l15) * GX_BK + pc8, xoff = GX_TILE + (wm * 64 + l15) * GX_BK + pc8;

    // per-tile DMA sources: uniform base pointers + this lane's constant byte offsets
    unsigned long long a_cur = 0, b_cur = 0, a_nxt = 0, b_nxt = 0;
    unsigned aoffb[2] = {0, 0}, boffb[2], aoffb_n[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) boffb[q] = (unsigned)((drow[q] * p.K + dcol[q]) * 2);
    const bool a_flat = p.a_div > p.M;                    // rows m -> A + m * a_inner (no division)
    auto arow = [&](int m) __attribute__((always_inline)) -> long {
        return a_flat ? (long)m * p.a_inner : (long)(m / p.a_div) * p.a_outer + (long)(m % p.a_div) * p.a_inner;
    };
    auto tile_src = [&](int rowtile, int c0, unsigned long long &ab, unsigned long long &bb, unsigned (&ao)[2]) __attribute__((always_inline)) {
        const int m0 = rowtile * 256;
        const int mb = m0 < p.M ? m0 : p.M - 1;
        const long base0 = arow(mb);
        ab = (unsigned long long)(p.A + base0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int am = m0 + drow[q];
            if (am >= p.M) am = p.M - 1;
            ao[q] = (unsigned)((arow(am) - base0 + dcol[q]) * 2);
        }
        bb = (unsigned long long)(p.B + (long)c0 * p.K);
    };
    auto issue = [&](int slot_, unsigned long long ab, unsigned long long bb, const unsigned (&ao)[2], int kslab) __attribute__((always_inline)) {
        const unsigned l = dma_lds + (unsigned)slot_ * (GX_STAGE * 2);
        ab += (unsigned)kslab * (GX_BK * 2);
        bb += (unsigned)kslab * (GX_BK * 2);
        asm volatile("" : "+s"(ab));
        asm volatile("" : "+s"(bb));
        if (DBG & 16) {   // ablation: weight pieces read CONTIGUOUS 1 KB (as a pre-tiled slab image would be): full 128-B lines
            const unsigned long long lin = bb - (unsigned)kslab * (GX_BK * 2) + (unsigned)kslab * (GX_TILE * 2) + (unsigned)(wave * 2) * 1024u + lane * 16u;
            gx_dma16((gx_ghalf_p)lin, l);
            gx_dma16((gx_ghalf_p)(lin + 1024), l + 1024);
        } else {
            gx_dma16((gx_ghalf_p)(bb + boffb[0]), l);
            gx_dma16((gx_ghalf_p)(bb + boffb[1]), l + 1024);
        }
        if (DBG & 32) {   // ablation: the same for the activation pieces
            const unsigned long long lin = ab - (unsigned)kslab * (GX_BK * 2) + (unsigned)kslab * (GX_TILE * 2) + (unsigned)(wave * 2) * 1024u + lane * 16u;
            gx_dma16((gx_ghalf_p)lin, l + GX_TILE * 2);
            gx_dma16((gx_ghalf_p)(lin + 1024), l + GX_TILE * 2 + 1024);
        } else {
            gx_dma16((gx_ghalf_p)(ab + ao[0]), l + GX_TILE * 2);
            gx_dma16((gx_ghalf_p)(ab + ao[1]), l + GX_TILE * 2 + 1024);
        }
    };

    int rowtile = 0, c0 = 0, rowtile_p = 0, c0_p = 0;
    tile_rc(0, rowtile, c0);
    tile_src(rowtile, c0, a_cur, b_cur, aoffb);
    issue(0, a_cur, b_cur, aoffb, 0);
    issue(1, a_cur, b_cur, aoffb, 1);

    float4_t acc[8][4];      // [weight tile g: columns 16 g ..][activation tile rt: rows 16 rt ..]
    half8_t wf[3], xa[4];
    // Epilogue of tile (e_rowtile, e_c0).  D row (reg r) = output column n = c0 + wn*128 + g*16 + 4 lq + r ;
    // D col (lane & 15) = output row m0 + wm*64 + rt*16 + l15.
    const bool o_flat = p.o_div > p.M;
    auto epilogue = [&](int e_rowtile, int e_c0, int e_zone) __attribute__((always_inline)) {
        if (DBG & 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) asm volatile("" ::"v"(acc[g][0]), "v"(acc[g][1]), "v"(acc[g][2]), "v"(acc[g][3]));
            return;
        }
        const int m0 = e_rowtile * 256;
        const int cw = e_c0 + wn * 128;                       // first column of this wave
        if (EPI == 1 && p.vT != nullptr && cw >= p.rope_cols) {
            // V third of the QKV projection: store TRANSPOSED, vT[n][h][d][t] (t contiguous).  Per weight tile g (16 d):
            // the four activation tiles (64 tokens) are gathered in the patch as [d][t], rows leave as 128 B
            const int nchunk = m0 / p.rope_T, t0 = m0 % p.rope_T + wm * 64;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) patch[(4 * lq + r) * GX_VLD + rt * 16 + l15] = (half_t)acc[g][rt][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int dd = (lane >> 3) + 8 * i, seg = lane & 7;
                    const half8_t v = *(LDSP(const half8_t))(patch + dd * GX_VLD + seg * 8);
                    const int cv = cw - p.rope_cols + g * 16 + dd;            // column inside V: h*64 + d
                    if (!(DBG & 1))
                        *(half8_t *)(p.vT + ((size_t)nchunk * (p.Ncols - p.rope_cols) + cv) * p.rope_T + t0 + seg * 8) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int mrow = m0 + wm * 64 + rt * 16;        // first of the 16 rows of this accumulator column block
            if (EPI == 1) {
                // rotary embedding (nn/TxModules.cpp:232-244): a head is 64 columns = tiles 4 hp .. 4 hp + 3; element d of
                // its first half (tiles 4 hp, 4 hp + 1) rotates with element d + 32 (tiles 4 hp + 2, 4 hp + 3): lane-local
                const float2 *tab = (const float2 *)p.rope + (size_t)((mrow + l15) % p.rope_T) * 32;
#pragma unroll
                for (int hp = 0; hp < 2; ++hp)
#pragma unroll
                    for (int gl = 0; gl < 2; ++gl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float2 cs = tab[gl * 16 + 4 * lq + r];
                            const float a = (float)(half_t)acc[4 * hp + gl][rt][r], b = (float)(half_t)acc[4 * hp + gl + 2][rt][r];
                            acc[4 * hp + gl][rt][r] = fmaf(cs.x, a, -(cs.y * b));
                            acc[4 * hp + gl + 2][rt][r] = fmaf(cs.y, a, cs.x * b);
                        }
            }
            // Rows leave through the wave's LDS patch as 4 rows x 256 B per store instruction (the wave's whole 128 columns
            // of a row).  tools/store_rate.hip: a 1 KB store instruction costs the CU ~27 cycles when contiguous, ~48 as 8
            // separate 128-B lines inside one 64 KB page and 90-130 when its rows lie 8 KB or more apart (a translation per
            // page) — with output rows of 3-8 KB, fewer and longer row segments per instruction is what counts.
            half_t *orow[2][2];
            bool ook[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = mrow + 8 * hh + (lane >> 4) + 4 * i;
                    ook[hh][i] = m < p.M;
                    const int mm = ook[hh][i] ? m : 0;
                    orow[hh][i] = p.out + (o_flat ? (long)mm * p.o_inner : (long)(mm / p.o_div) * p.o_outer + (long)(mm % p.o_div) * p.o_inner);
                }
            half4_t hv[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float4_t v = acc[g][rt];
                if (EPI == 0) {
                    if (p.bias != nullptr) {
                        const float4_t bv = *(LDSP(const float4_t))((LDSP(const float))(smem3 + GX_OFF_BIAS) + e_zone * 256 + wn * 128 + g * 16 + 4 * lq);
                        v += bv;
                    }
                    // the activation code is a wave-uniform switch outside the element loop
                    if (p.act == 3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = 5.0f * fast_tanh(v[r]);
                    } else if (p.act == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], 0);
                    } else if (p.act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], 1);
                    } else if (p.act == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], 2);
                    }
                }
                hv[g] = __builtin_convertvector(v, half4_t);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                 // rows 8 hh .. 8 hh + 7 of the accumulator tile
                if ((l15 >> 3) == hh) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) *(LDSP(half4_t))(patch + (l15 & 7) * GX_PLD + g * 16 + 4 * lq) = hv[g];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int prow = (lane >> 4) + 4 * i, seg = lane & 15;
                    const half8_t v = *(LDSP(const half8_t))(patch + prow * GX_PLD + seg * 8);
                    if (DBG & 1) {
                        asm volatile("" ::"v"(v));
                    } else if (ook[hh][i]) {
                        // streaming (nt) stores: the output is 4-56 GB per launch and is not read back by this kernel; written with
                        // the default policy it evicts the L2-resident weight slice and the shared activation panels (measured on
                        // random operands, same box: sup head 56.7 -> 55.1 ms, transformer CRF 10.98 -> 9.82 ms; DBG 4 = default policy)
                        if (DBG & 4) *(half8_t *)(orow[hh][i] + cw + seg * 8) = v;
                        else __builtin_nontemporal_store(v, (half8_t *)(orow[hh][i] + cw + seg * 8));
                    }
                }
                __builtin_amdgcn_wave_barrier();
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

        gx_static_for<KS>([&](auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int slot_ = ks & 3;
            // (KS % 4 == 0: slab g of the stream sits in ring slot ks & 3 in every tile)
            if (!grpB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // B1
            asm volatile("" ::: "memory");
            {
                LDSP(const half_t) sp = stage + slot_ * GX_STAGE;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) xa[rt] = *(LDSP(const half8_t))(sp + xoff + rt * 16 * GX_BK);
#pragma unroll
                for (int g = 0; g < 3; ++g) wf[g] = *(LDSP(const half8_t))(sp + woff + g * 16 * GX_BK);
                constexpr int kt = ks + 2;
                if (kt < KS) issue(kt & 3, a_cur, b_cur, aoffb, kt);
                else issue(kt & 3, a_nxt, b_nxt, aoffb_n, kt - KS);
                // group A's epilogue of the PREVIOUS tile sits here, behind L(0) of this tile: it then runs beside
                // group B's M(KS-1) + epilogue of that tile (half a slab later by construction) instead of before it
                if (ks == 0 && !grpB && ti > 0) epilogue(rowtile_p, c0_p, (ti - 1) & 1);
                // this tile's bias values -> LDS zone ti & 1 (one extra DMA of wave 0: the counted waits only get stricter)
                if (EPI == 0 && ks == 1 && wave == 0 && p.bias != nullptr)
                    gx_dma16f(p.bias + c0 + lane * 4, lds0 + GX_OFF_BIAS + (unsigned)(ti & 1) * 1024u);
            }
            if (grpB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // B2
            asm volatile("" ::: "memory");
            {
                if (ks == 0) {
#pragma unroll
                    for (int g = 0; g < 8; ++g)
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) acc[g][rt] = (float4_t)(0.0f);
                }
                LDSP(const half_t) sp = stage + slot_ * GX_STAGE;
                __builtin_amdgcn_s_setprio(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (!(DBG & 8)) {
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) acc[g][rt] = gx_mfma(wf[g % 3], xa[rt], acc[g][rt]);
                    } else {
                        asm volatile("" ::"v"(wf[g % 3]), "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 3 < 8) wf[g % 3] = *(LDSP(const half8_t))(sp + woff + (g + 3) * 16 * GX_BK);
                    __builtin_amdgcn_sched_barrier(0);
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

// Column tiles per group: all of them while the weight matrix fits an XCD's L2 beside the activation stream (<= 2 MB),
// else the largest power-of-two count whose weight slice does (and that leaves 1, 2, 4 or 8 groups).
static int gx_col_group(const GemmArgs *a) {
    const int ncol = a->Ncols / 256;
    const size_t tile_bytes = (size_t)256 * a->K * 2;
    if ((size_t)ncol * tile_bytes <= (size_t)(2u << 20)) return ncol;
    for (int cg = ncol; cg >= 1; --cg) {
        if (ncol % cg != 0) continue;
        const int ngrp = ncol / cg;
        if (ngrp != 1 && ngrp != 2 && ngrp != 4 && ngrp != 8) continue;
        if ((size_t)cg * tile_bytes <= (size_t)(2u << 20)) return cg;
    }
    return ncol;
}

// 0 = launched; 1 = shape / epilogue not covered (caller uses gemm256_kernel or gemm_dma_kernel).
extern "C" int mibc_launch_gemm256x(hipStream_t s, const GemmArgs *a) {
    // M >= 256 (round 6; was 2048): every batch size that fills one row tile takes THIS kernel, so a chunk's scores no longer depend on
    // whether it was called alone (M = 1024 / 1666 rows: 32x32x16 kernels until round 5, a different summation tree) or in a batch
    if (a->Ncols % 256 != 0 || a->M < 256 || a->ncols_valid != 0) return 1;
    if (a->K != 512 && a->K != 1024) return 1;
    if (a->epi_mode != 0 && a->epi_mode != 1) return 1;
    if (a->epi_mode == 1 && (a->rope_T % 256 != 0 || a->rope_cols % 128 != 0 || a->vT == nullptr)) return 1;
    int cg = gx_col_group(a);
    int kdbg = 0;
    int stagger = 0;
#ifdef MIBC_DEBUG_KERNELS
    stagger = MIBC_ENV_INT("MIBC_GX_STAGGER", 0);
    if (MIBC_ENV_INT("MIBC_GX_OFF", 0)) return 1;          // A/B inside the engine: fall back to gemm256_kernel
    {
        const int c = MIBC_ENV_INT("MIBC_GX_CG", 0), ncol = a->Ncols / 256;
        if (c > 0 && ncol % c == 0 && (ncol / c == 1 || ncol / c == 2 || ncol / c == 4 || ncol / c == 8)) cg = c;
    }
#endif
#ifdef MIBC_DEBUG_KERNELS
    // microbenchmark switches (tools/gemm_bench.py): dbg = 0x2000 | (col group << 4) | ablation bits (1, 2, 8)
    if (a->dbg != 0 && (a->dbg & 0xf000) != 0x2000) return 1;
    if (a->dbg != 0) {
        kdbg = (a->dbg & 0xf) | ((a->dbg >> 6) & 0x30);      // 0x400 / 0x800 -> ablations 16 / 32 (bits 4-9: column group)
        const int c = (a->dbg >> 4) & 0x3f;
        const int ncol = a->Ncols / 256;
        if (c > 0 && ncol % c == 0 && (ncol / c == 1 || ncol / c == 2 || ncol / c == 4 || ncol / c == 8)) cg = c;
    }
#else
    if (a->dbg != 0) return 1;
#endif
    const int ncu = mibc_ncu();   // of the launching thread's current device
    const long ntiles = (long)((a->M + 255) / 256) * (a->Ncols / 256);
    int grid = (ncu / 8) * 8;
    if (ntiles < grid) grid = (int)((ntiles + 7) / 8) * 8;
#define GX_LAUNCH(KS_, E_, D_)                                                                                   \
    do {                                                                                                         \
        MIBC_LDS_ATTR_ONCE((gemm256x_kernel<KS_, E_, D_>), GX_LDS_BYTES);                                        \
        hipLaunchKernelGGL((gemm256x_kernel<KS_, E_, D_>), dim3(grid), dim3(512), GX_LDS_BYTES, s, *a, cg, stagger);      \
        return 0;                                                                                                \
    } while (0)
#ifdef MIBC_DEBUG_KERNELS
    if (kdbg != 0 && a->epi_mode == 0) {
        if (a->K == 512) {
            if (kdbg == 1) GX_LAUNCH(16, 0, 1);
            if (kdbg == 4) GX_LAUNCH(16, 0, 4);
            if (kdbg == 2) GX_LAUNCH(16, 0, 2);
            if (kdbg == 8) GX_LAUNCH(16, 0, 8);
            if (kdbg == 10) GX_LAUNCH(16, 0, 10);
        } else {
            if (kdbg == 1) GX_LAUNCH(32, 0, 1);
            if (kdbg == 4) GX_LAUNCH(32, 0, 4);
            if (kdbg == 2) GX_LAUNCH(32, 0, 2);
            if (kdbg == 8) GX_LAUNCH(32, 0, 8);
            if (kdbg == 16) GX_LAUNCH(32, 0, 16);
            if (kdbg == 32) GX_LAUNCH(32, 0, 32);
            if (kdbg == 48) GX_LAUNCH(32, 0, 48);
            if (kdbg == 50) GX_LAUNCH(32, 0, 50);
            if (kdbg == 56) GX_LAUNCH(32, 0, 56);
        }
        return 1;
    }
#endif
    if (a->epi_mode == 1) {
        if (a->K == 512) GX_LAUNCH(16, 1, 0);
        GX_LAUNCH(32, 1, 0);
    }
    if (a->K == 512) GX_LAUNCH(16, 0, 0);
    GX_LAUNCH(32, 0, 0);
#undef GX_LAUNCH
}
