// dorado_amd/csrc/txlayer.hip — one launch for everything of a transformer encoder layer that follows the attention
// (sup@v5, d_model 512):
//      t  = attn . Wo^T + bo                      out-proj              nn/TxModules.cpp:653-661 ("OUTP")
//      x1 = RMSNorm(t + alpha x) * n1             residual norm 1       :662-678 ("LNORM1"), nn/RMSNorm.cpp:14-18
//      h  = silu(gate) * y,  [y | gate] = x1 . W1^T   FC1 + SwiGLU      :679-694 ("FC1+SILU"), :133-176 (GatedMLP)
//      u  = h . W2^T                              FC2                   :695-708 ("FC2")
//      x  = RMSNorm(u + alpha x1) * n2            residual norm 2       :709-723 ("LNORM2")
// replacing five launches (gemm256 out-proj, residual_rmsnorm, gemm256 FC1+SwiGLU, gemm256 FC2, residual_rmsnorm) that
// moved 29 GB of HBM traffic per layer at 1 M tokens (the [tokens][2048] SwiGLU tensor alone: 4.3 GB out + 4.3 GB in)
// with one that reads attn and x and writes x: 3 GB.
//
// Machine mapping ("128-row tile resident, FF walked in slabs, FC2 accumulator in registers"):
//   * persistent, one workgroup of FOUR waves per CU — one wave per SIMD with the whole 512-register file: wave w owns 32
//     token rows of the 128-row tile for the whole layer tail:
//        - its rows of the FC1 input (attn, then x1) sit in 128 registers as the 32 B-operand fragments of
//          v_mfma_f32_32x32x16_f16 (tokens are the MFMA's column dimension, weights its row dimension, as in gemm256.hip);
//        - the 32 x 512 f32 result tile of the out-proj / of FC2 is 16 accumulator tiles = 256 registers (AGPRs);
//        - FC1 is walked in slabs of 32 hidden units (64 weight rows: 32 y + 32 gate); its two 32 x 32 accumulators hold,
//          after SwiGLU and f16 packing, EXACTLY the B-operand fragment of the FC2 MFMAs for those 32 hidden units (the
//          k index inside a 16-group is permuted — 0-3, 8-11 | 4-7, 12-15 — and W2's image is stored with the same
//          permutation): the [tokens][FF] intermediate never leaves the register file, not even to LDS;
//   * weights are the only operand that moves: the layer's 6.5 MB stream through a 4-slot LDS ring of 32 KB stages by direct
//     LDS DMA (global_load_lds_dwordx4), in CONSUMPTION order and in MFMA-fragment order (the host lays the image out:
//     every ds_read_b128 is lane-linear, conflict-free, no swizzle), three stages ahead, counted vmcnt + one raw barrier
//     per stage (= 32 MFMAs per wave); every stage is read by all four waves (LDS fragment traffic: 1 KB per MFMA and
//     wave = 128 B/clk/CU at the matrix pipe's peak, half of what the 256 x 256 GEMM tile needs per flop once its
//     activation operand is counted);
//   * SwiGLU of slab j runs inside the FC2 stage of slab j - 1 (VALU beside MFMAs of the same wave);
//   * both residual RMSNorms run on the accumulators: 8 rows at a time go through a per-wave 8 KB LDS patch (XOR-swizzled
//     16-byte slots) into row-per-wave form, where a lane owns the same 8 columns as in residual_rmsnorm_kernel and the
//     reduction is the same xor tree — phase A (out-proj + norm 1) is bit-identical to the unfused kernels; FC2 sums the
//     same products in a different order inside each 16-group, so x differs from the unfused path by f32 rounding only.
#include "common.h"
#include "cluster_util.h"
#include "engine.h"

#include <type_traits>
#include <vector>

#define TL_STAGE_BYTES 32768
#define TL_NS 4
#define TL_OFF_SIDE (TL_NS * TL_STAGE_BYTES)           // 4 waves x 8 KB
#define TL_LDS_BYTES (TL_OFF_SIDE + 4 * 8192)
#define TL_D 512
#define TL_FD 8                      // fragment look-ahead in MFMAs (8 x 32 cycles of LDS latency cover)

// FC1's two accumulator tiles must live in VGPRs: the 16 tiles of the FC2 / out-proj result fill the AGPR half of the
// register file exactly, and left to itself hipcc puts EVERY MFMA result into AGPRs (288 > 256: it then swaps result tiles
// and spills input fragments inside the loop).  Inline asm with "v" constraints pins them.  hipcc's hazard recogniser
// does not see asm MFMAs; the two hazards that exist are handled where they arise (tl_fc1_results_ready).
__device__ __forceinline__ void tl_mfma_v_first(float16_t &c, half8_t a, half8_t b) {      // c = a . b
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tl_mfma_v(float16_t &c, half8_t a, half8_t b) {            // c += a . b
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// Placed two (builtin) MFMAs behind the last asm MFMA of a slab, pinned by scheduling barriers: every later read of the
// FC1 accumulators depends on this statement, and two 8-pass MFMA issues (>= 64 cycles) cover the 11 wait states an
// "XDL write VGPR -> VALU read" needs.
__device__ __forceinline__ void tl_fc1_results_ready(float16_t &y, float16_t &g) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(y), "+v"(g));
    __builtin_amdgcn_sched_barrier(0);
}

struct TxLayerArgs {
    const half_t *attn;   // [R][512]
    half_t *x;            // [R][512] in: layer input (residual) / out: layer output
    const half_t *wimg;   // weight image, 32 KB stages in consumption order (tx_layer_image)
    const float *bo, *n1, *n2;
    float alpha;
    long R;
    int FF;
};

// MODE 3 = whole tail; 1 = out-proj + norm 1 only (x <- x1); 2 = MLP + norm 2 only (x holds x1)   [test decomposition]
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tx_layer_kernel(TxLayerArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDSP(unsigned char) smem3 = (LDSP(unsigned char))smem;
    const unsigned lds0 = (unsigned)(size_t)smem3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int NJ = p.FF >> 5;
    const int first_stage = (MODE & 1) ? 0 : 16;
    const int stages_per_tile = ((MODE & 1) ? 16 : 0) + ((MODE & 2) ? 3 * NJ : 0);

    const long ntiles = (p.R + 127) >> 7;
    long my_tiles = 0;
    if ((long)blockIdx.x < ntiles) my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    if (my_tiles == 0) return;

    // ---- weight stream: stage g of this workgroup = stage (g % stages_per_tile) of the image -> ring slot g & 3 ----
    const unsigned long long wbase = (unsigned long long)p.wimg + (unsigned long long)first_stage * TL_STAGE_BYTES +
                                     (unsigned)(wave * 8 * 1024 + lane * 16);
    const unsigned dma_dst = lds0 + (unsigned)(wave * 8 * 1024);
    int g_issue = 0;        // next stage to request (ring slot g_issue & 3)
    int st_issue = 0;       // g_issue % stages_per_tile: stage of the image (past the last tile the stream simply wraps:
                            // two stages nobody reads — the waits are counted, so the request count per stage is constant)
    auto issue_stage = [&]() __attribute__((always_inline)) {
        const unsigned long long src = wbase + (unsigned long long)(unsigned)st_issue * TL_STAGE_BYTES;
        const unsigned dst = dma_dst + (unsigned)(g_issue & 3) * TL_STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 8; ++q) cl_dma16((ghalf_p)(src + q * 1024), dst + q * 1024);
        ++g_issue;
        st_issue = (st_issue + 1 == stages_per_tile) ? 0 : st_issue + 1;
    };
    // ---- stage protocol.  Boundary B(s) (executed inside stage s - 1, TL_FD MFMAs before its end; B(0) up front):
    //   wait until my requests for stage s have landed (the 8 of stage s + 1 stay in flight), barrier (everybody's have;
    //   and everybody has issued — and consumed — every read of stage s - 2, whose slot the next request overwrites),
    //   request stage s + 2.  The fragment ring fr[] runs TL_FD MFMAs ahead of the matrix pipe, across stage borders.
    int g_use = 0;          // stage whose fragments are being requested from LDS next
    LDSP(const half8_t) fcur = (LDSP(const half8_t))smem3 + lane;
    LDSP(const half8_t) fnext = fcur;
    auto boundary = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fnext = (LDSP(const half8_t))(smem3 + (unsigned)(g_use & 3) * TL_STAGE_BYTES) + lane;
        issue_stage();
        ++g_use;
    };
    half8_t fr[TL_FD];
    issue_stage();
    issue_stage();
    boundary();
    fcur = fnext;
#pragma unroll
    for (int i = 0; i < TL_FD; ++i) fr[i] = fcur[i * 64];
    // one stage = 32 fragments, consumed in order by 32 MFMAs: mf(i, fragment i)
    auto run_stage = [&](auto &&mf) __attribute__((always_inline)) {
        cl_static_for<32>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            const half8_t a = fr[i % TL_FD];
            if (i == 32 - TL_FD) boundary();
            fr[i % TL_FD] = (i + TL_FD < 32) ? fcur[(i + TL_FD) * 64] : fnext[(i + TL_FD - 32) * 64];
            mf(i_c, a);
        });
        fcur = fnext;
    };

    half8_t xf[32];          // this wave's 32 token rows x 512 as B fragments: xf[ks] = row l31, k = 16 ks + 8 lhi + 0..7
    float16_t out[16];       // 32 x 512 f32: out[c][r] = column 32 c + (r&3) + 8 (r>>2) + 4 lhi of token row l31
    LDSP(unsigned char) side = smem3 + TL_OFF_SIDE + wave * 8192;
    // out[c] += W-fragments of one 32 KB stage (2 k-steps x 16 column tiles) . (b0, b1)
    auto stage_512 = [&](half8_t b0, half8_t b1) __attribute__((always_inline)) {
        run_stage([&](auto i_c, half8_t a) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            out[i & 15] = mfma32x32x16(a, (i < 16) ? b0 : b1, out[i & 15]);
        });
    };

    // residual RMSNorm of this wave's 32 rows on the accumulators, 8 rows per pass through the wave's LDS patch.
    //   v = f16(out [+ bias]) + alpha * res ;  y = f16((v * rsqrt(mean(v^2) + eps)) * w)
    // RES_FROM_XF: the residual rows are the xf fragments (x1), else they are read from p.x.
    // TO_XF: the result becomes the new xf fragments (and is not stored), else it is stored to p.x.
    auto norm_rows = [&](long row0, const float *bias, const float *wn, auto res_from_xf, auto to_xf) __attribute__((always_inline)) {
        constexpr bool RES_FROM_XF = decltype(res_from_xf)::value, TO_XF = decltype(to_xf)::value;
        float w[8];   // norm weights of this lane's 8 columns in row-per-wave form (columns 8 lane .. 8 lane + 7)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = wn[lane * 8 + e];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const bool mine = (l31 >> 3) == ps;
            const int i_own = l31 & 7;
            half8_t res[8];
            if (RES_FROM_XF) {
                if (mine) {
#pragma unroll
                    for (int ks = 0; ks < 32; ++ks)
                        *(LDSP(half8_t))(side + i_own * 1024 + (((2 * ks + lhi) ^ i_own) << 4)) = xf[ks];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) res[i] = *(LDSP(const half8_t))(side + i * 1024 + ((lane ^ i) << 4));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    long r = row0 + 8 * ps + i;
                    if (r >= p.R) r = p.R - 1;
                    res[i] = *(const half8_t *)(p.x + r * TL_D + lane * 8);
                }
            }
            if (mine) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        float4_t bv = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (bias != nullptr) bv = *(const float4_t *)(bias + 32 * c + 8 * rq + 4 * lhi);
                        half4_t q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = (half_t)(out[c][4 * rq + e] + bv[e]);
                        // logical byte column (32 c + 8 rq + 4 lhi) * 2 -> 16-byte slot 4 c + rq, offset 8 lhi
                        *(LDSP(half4_t))(side + i_own * 1024 + (((4 * c + rq) ^ i_own) << 4) + 8 * lhi) = q;
                    }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const half8_t a = *(LDSP(const half8_t))(side + i * 1024 + ((lane ^ i) << 4));
                float v[8];
                float ss = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = (float)a[e] + (float)res[i][e] * p.alpha;
                    ss += v[e] * v[e];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                const float rstd = rsqrtf(ss / (float)TL_D + 1e-5f);
                half8_t y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (half_t)((v[e] * rstd) * w[e]);
                if (TO_XF) {
                    *(LDSP(half8_t))(side + i * 1024 + ((lane ^ i) << 4)) = y;
                } else {
                    const long r = row0 + 8 * ps + i;
                    if (r < p.R) *(half8_t *)(p.x + r * TL_D + lane * 8) = y;
                }
            }
            if (TO_XF && mine) {
#pragma unroll
                for (int ks = 0; ks < 32; ++ks)
                    xf[ks] = *(LDSP(const half8_t))(side + i_own * 1024 + (((2 * ks + lhi) ^ i_own) << 4));
            }
        }
    };

    for (long ti = 0; ti < my_tiles; ++ti) {
        const long tile = blockIdx.x + ti * (long)gridDim.x;
        const long row0 = tile * 128 + wave * 32;       // first row of this wave
        {   // B fragments of the tile's input rows (attn, or x1 in the MLP-only mode)
            long r = row0 + l31;
            if (r >= p.R) r = p.R - 1;
            const half_t *src = ((MODE & 1) ? p.attn : (const half_t *)p.x) + r * TL_D + 8 * lhi;
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) xf[ks] = *(const half8_t *)(src + 16 * ks);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[c][r] = 0.0f;

        if (MODE & 1) {
            // ---- phase A: out-proj, 16 stages of 32 k ----
            cl_static_for<16>([&](auto st_c) __attribute__((always_inline)) {
                constexpr int st = decltype(st_c)::value;
                stage_512(xf[2 * st], xf[2 * st + 1]);
            });
            if (MODE == 1) {
                norm_rows(row0, p.bo, p.n1, std::false_type{}, std::false_type{});
                continue;
            }
            norm_rows(row0, p.bo, p.n1, std::false_type{}, std::true_type{});     // xf <- x1
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[c][r] = 0.0f;
        }

        // ---- phase B: gated MLP, slabs of 32 hidden units ----
        float16_t ay, ag;
        half8_t ha[2], hb[2];
        // FC1 of one slab: two stages of 256 k each
        auto fc1_slab = [&]() __attribute__((always_inline)) {
            cl_static_for<2>([&](auto h_c) __attribute__((always_inline)) {
                constexpr int h = decltype(h_c)::value;
                run_stage([&](auto i_c, half8_t a) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_c)::value;      // fragment (s = i / 2, t = i % 2)
                    if (h == 0 && i == 0) tl_mfma_v_first(ay, a, xf[0]);
                    else if (h == 0 && i == 1) tl_mfma_v_first(ag, a, xf[0]);
                    else if ((i & 1) == 0) tl_mfma_v(ay, a, xf[16 * h + (i >> 1)]);
                    else tl_mfma_v(ag, a, xf[16 * h + (i >> 1)]);
                });
            });
        };
        // SwiGLU of the finished slab (nn/TxModules.cpp:171-175 on the f16-rounded FC1 outputs) -> FC2 B fragments
        auto swiglu_elem = [&](half8_t (&hn)[2], int r) __attribute__((always_inline)) {
            const float y = (float)(half_t)ay[r], gt = (float)(half_t)ag[r];
            hn[r >> 3][r & 7] = (half_t)(gt * fast_sigmoid(gt) * y);
        };
        // FC2 stage of the previous slab (fragments hp) with the SwiGLU of the current one (-> hn) between its MFMAs
        auto fc2_stage = [&](const half8_t (&hp)[2], half8_t (&hn)[2], auto with_swiglu) __attribute__((always_inline)) {
            run_stage([&](auto i_c, half8_t a) __attribute__((always_inline)) {
                constexpr int i = decltype(i_c)::value;
                out[i & 15] = mfma32x32x16(a, hp[i >> 4], out[i & 15]);
                if (decltype(with_swiglu)::value) {
                    // element e behind MFMA 2 e + 2 (e < 15), the last one behind MFMA 31
                    if (i == 1) tl_fc1_results_ready(ay, ag);
                    if (i >= 2 && (i & 1) == 0) swiglu_elem(hn, (i >> 1) - 1);
                    if (i == 31) swiglu_elem(hn, 15);
                }
            });
        };
        // slab 0: FC1, SwiGLU alone
        fc1_slab();
        asm volatile("s_nop 7\n\ts_nop 3" : "+v"(ay), "+v"(ag));      // XDL write -> VALU read: 11 wait states (asm MFMAs)
#pragma unroll
        for (int r = 0; r < 16; ++r) swiglu_elem(ha, r);
        // slabs 1 .. NJ-1 in pairs (static fragment registers): odd slab -> hb, even slab -> ha
        for (int j = 1; j + 1 < NJ; j += 2) {
            fc1_slab();
            fc2_stage(ha, hb, std::true_type{});
            fc1_slab();
            fc2_stage(hb, ha, std::true_type{});
        }
        fc1_slab();                    // slab NJ-1 (odd)
        fc2_stage(ha, hb, std::true_type{});       // FC2 of slab NJ-2, SwiGLU of slab NJ-1
        fc2_stage(hb, ha, std::false_type{});      // FC2 of slab NJ-1

        // residual = x1 = this wave's input fragments of the MLP (still in xf)
        norm_rows(row0, nullptr, p.n2, std::true_type{}, std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight image of one layer: 32 KB stages = 32 fragments of [64 lanes][8 halfs], in consumption order:
//   16 x Wo(st) | W1(0,0) W1(0,1) | W1(1,0) W1(1,1) W2(0) | W1(2,0) W1(2,1) W2(1) | ... | W1(NJ-1,0) W1(NJ-1,1) W2(NJ-2) | W2(NJ-1)
// Wo(st): fragment (s, c) = Wo[32 c + l31][32 st + 16 s + 8 lhi + e]
// W1(j,h): fragment (s, t) = W1[(t ? FF : 0) + 32 j + l31][256 h + 16 s + 8 lhi + e]          (y rows, then gate rows)
// W2(j):  fragment (s, c) = W2[32 c + l31][32 j + 16 s + 4 lhi + (e & 3) + 8 (e >> 2)]        (k permuted: see the header)
// wo [512][512], w1 [2 FF][512], w2 [512][FF] row-major f32 (module.parameters() layout).
std::vector<half_t> tx_layer_image(const float *wo, const float *w1, const float *w2, int FF) {
    const int NJ = FF / 32, C = TL_D;
    const size_t stage_halfs = TL_STAGE_BYTES / 2;
    std::vector<half_t> img((size_t)(16 + 3 * NJ) * stage_halfs);
    size_t st_idx = 0;
    auto frag = [&](size_t stage, int f) { return img.data() + stage * stage_halfs + (size_t)f * 512; };
    for (int st = 0; st < 16; ++st, ++st_idx)
        for (int s = 0; s < 2; ++s)
            for (int c = 0; c < 16; ++c) {
                half_t *d = frag(st_idx, s * 16 + c);
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e)
                        d[lane * 8 + e] = (half_t)wo[(size_t)(32 * c + (lane & 31)) * C + 32 * st + 16 * s + 8 * (lane >> 5) + e];
            }
    auto put_w1 = [&](int j) {
        for (int h = 0; h < 2; ++h, ++st_idx)
            for (int s = 0; s < 16; ++s)
                for (int t = 0; t < 2; ++t) {
                    half_t *d = frag(st_idx, 2 * s + t);
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e)
                            d[lane * 8 + e] = (half_t)w1[(size_t)((t ? FF : 0) + 32 * j + (lane & 31)) * C + 256 * h + 16 * s +
                                                         8 * (lane >> 5) + e];
                }
    };
    auto put_w2 = [&](int j) {
        for (int s = 0; s < 2; ++s)
            for (int c = 0; c < 16; ++c) {
                half_t *d = frag(st_idx, s * 16 + c);
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e)
                        d[lane * 8 + e] = (half_t)w2[(size_t)(32 * c + (lane & 31)) * FF + 32 * j + 16 * s + 4 * (lane >> 5) +
                                                     (e & 3) + 8 * (e >> 2)];
            }
        ++st_idx;
    };
    put_w1(0);
    for (int j = 1; j < NJ; ++j) {
        put_w1(j);
        put_w2(j - 1);
    }
    put_w2(NJ - 1);
    return img;
}

bool tx_layer_supported(int d_model, int ff) { return d_model == TL_D && ff >= 128 && ff % 64 == 0; }

// mode: 3 whole layer tail, 1 out-proj + norm 1 only, 2 MLP + norm 2 only (tests).  0 = launched, 1 = shape not covered.
extern "C" int mibc_launch_tx_layer(hipStream_t s, const half_t *attn, half_t *x, const half_t *wimg, const float *bo,
                                    const float *n1, const float *n2, float alpha, long R, int FF, int mode) {
    if (!tx_layer_supported(TL_D, FF) || R <= 0) return 1;
    TxLayerArgs a{attn, x, wimg, bo, n1, n2, alpha, R, FF};
    const long ntiles = (R + 127) / 128;
    long grid = mibc_ncu();
    if (ntiles < grid) grid = ntiles;
#define TL_LAUNCH(M_)                                                                                   \
    do {                                                                                                \
        MIBC_LDS_ATTR_ONCE((tx_layer_kernel<M_>), TL_LDS_BYTES);                                        \
        hipLaunchKernelGGL((tx_layer_kernel<M_>), dim3((unsigned)grid), dim3(256), TL_LDS_BYTES, s, a); \
    } while (0)
    if (mode == 1) TL_LAUNCH(1);
    else if (mode == 2) TL_LAUNCH(2);
    else TL_LAUNCH(3);
#undef TL_LAUNCH
    return 0;
}
