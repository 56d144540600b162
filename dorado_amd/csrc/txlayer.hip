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
//     reduction is the same xor tree; the out-proj adds its bias first instead of last and FC2 sums the same products in a
//     different order inside each 16-group, so x differs from the unfused path by f32 rounding only.
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
#define TL_PITCH 1056                // bytes per row of the epilogue patch (1024 + 32: conflict-free quads, see norm_rows)
#define TL_FD 4                      // fragment look-ahead in MFMAs (8 x 32 cycles of LDS latency cover)

// FC1's two accumulator tiles must live in VGPRs: the 16 tiles of the FC2 / out-proj result fill the AGPR half of the
// register file exactly, and left to itself hipcc puts EVERY MFMA result into AGPRs (288 > 256: it then swaps result tiles
// and spills input fragments inside the loop; pinning builtin MFMAs with empty "+v" asm statements makes it compute in
// AGPRs and copy: ~5 v_accvgpr moves per MFMA).  Inline asm with "v" constraints pins them for real.  hipcc's hazard
// recogniser does not see asm MFMAs: the LAST asm MFMA of each accumulator chain carries the wait states an
// "XDL write VGPR -> VALU read" needs (19 for a 16-pass MFMA) inside its own asm statement, so whatever hipcc places behind
// it (SwiGLU, register copies) is safe (8-pass XDL: 12 states).  Chain-internal dependencies (same vDst as SrcC, one MFMA
// apart) need none.  The other hazard — a VGPR written by a VALU instruction (a compiler v_mov forming an operand tuple)
// read by the MFMA within two states (cdna_hip_programming.md §5.7 item 2) — must not occur: an s_nop in front of every
// statement costs ~50 cycles per MFMA on a two-chain accumulation (it lands between MFMAs on the same accumulator), so
// instead tools/check_asm_hazards.py scans the compiled kernel for it (tests/test_host_cpu.py runs the scan).
__device__ __forceinline__ void tl_mfma_v_first(float16_t &c, half8_t a, half8_t b) {      // c = a . b
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tl_mfma_v(float16_t &c, half8_t a, half8_t b) {            // c += a . b
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tl_mfma_v_last(float16_t &c, half8_t a, half8_t b) {       // c += a . b, then safe to read
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 11" : "+v"(c) : "v"(a), "v"(b));
}

struct TxLayerArgs {
    const half_t *attn;   // [R][512]
    half_t *x;            // [R][512] in: layer input (residual) / out: layer output
    const half_t *wimg;   // weight image, 32 KB stages in consumption order (tx_layer_image)
    const float *bo, *n1, *n2;
    float alpha;
    long R;
    int FF;
    // DBG template parameter = timing ablations (wrong results; instantiated for MODE 2 only): 1 = every request fetches
    // stage 0 (always L2-hot), 2 = no requests at all, 4 = no epilogues (no norm, no stores), 8 = no input-fragment loads
};

// MODE 3 = whole tail; 1 = out-proj + norm 1 only (x <- x1); 2 = MLP + norm 2 only (x holds x1); 6 = MLP only, x <- the raw
// FC2 result   [test decomposition]
template <int MODE, int DBG = 0>
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
        const unsigned long long src = wbase + (unsigned long long)(unsigned)((DBG & 1) ? 0 : st_issue) * TL_STAGE_BYTES;
        const unsigned dst = dma_dst + (unsigned)(g_issue & 3) * TL_STAGE_BYTES;
        if (!(DBG & 2)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cl_dma16((ghalf_p)(src + q * 1024), dst + q * 1024);
        }
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
    // rows of attn / x through buffer descriptors: address = descriptor + SGPR row offset + one lane-constant VGPR offset
    // (+ immediate), rows past R read as zero and their stores are dropped by the bounds check — no per-row address
    // registers, no tail branches
    const unsigned row_bytes = TL_D * 2;
    const __amdgpu_buffer_rsrc_t rs_attn = __builtin_amdgcn_make_buffer_rsrc((void *)p.attn, 0, (int)(p.R * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)(p.R * row_bytes), 0x00020000);
    const int voff_row = lane * 16;                       // row-per-wave form: columns 8 lane .. 8 lane + 7
    const int voff_frag = l31 * (int)row_bytes + 16 * lhi;   // fragment form: token row l31, k = 8 lhi (+ 16 ks)
    // out[c] += W-fragments of one 32 KB stage (2 k-steps x 16 column tiles) . (b0, b1)
    auto stage_512 = [&](half8_t b0, half8_t b1) __attribute__((always_inline)) {
        run_stage([&](auto i_c, half8_t a) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            out[i & 15] = mfma32x32x16(a, (i < 16) ? b0 : b1, out[i & 15]);
        });
    };

    // residual RMSNorm of this wave's 32 rows on the accumulators: 4 rows per pass through the wave's LDS patch (rows of
    // TL_PITCH = 1056 bytes: every access below is base + immediate, and the 16 lanes of a pass that write 8-byte quads hit
    // 16 different bank groups), then row-per-wave form: a lane owns columns 8 lane .. 8 lane + 7 as in
    // residual_rmsnorm_kernel and the reduction is the same xor tree.
    //   v = f16(out) + alpha * res ;  y = f16((v * rsqrt(mean(v^2) + eps)) * w) -> p.x
    // The 32 residual rows (res[], load_res) are already in registers: beside LDS-DMA traffic hipcc drains vmcnt(0) for every
    // VGPR-returning load, so loads inside the passes — or register spills, which are scratch loads — would each cost a
    // full memory round trip behind the stores of the pass before.
    half8_t res[32];   // residual rows of this wave in row-per-wave form (lane: columns 8 lane .. 8 lane + 7), see load_res
    // fresh: rows this wave stored earlier in the launch (x1) -> sc1: served by L2, never by a stale L1 line.
    // (Requesting the rows earlier — under the last MFMA stages, into registers the input fragments no longer need — was
    // tried: hipcc then parks fragments in accumulator registers and reads them back right in front of the asm MFMAs, the
    // hazard tools/check_asm_hazards.py rejects.)
    auto load_res = [&](auto r_c, long row0, auto fresh) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value;
        res[r] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_row, (int)(row0 * row_bytes) + r * (int)row_bytes,
                                                                                  decltype(fresh)::value ? 16 : 0));
    };
    auto norm_rows = [&](long row0, const float *wn, auto fresh) __attribute__((always_inline)) {
        if (DBG & 4) return;
        float w[8];   // norm weights of this lane's 8 columns
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = wn[lane * 8 + e];
        const int soff0 = (int)(row0 * row_bytes);
        cl_static_for<32>([&](auto r_c) __attribute__((always_inline)) { load_res(r_c, row0, fresh); });
        const int i_own = l31 & 3;
        LDSP(unsigned char) own_row = side + i_own * TL_PITCH + 8 * lhi;
        LDSP(const unsigned char) my_slot = side + lane * 16;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            if ((l31 >> 2) == ps) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        half4_t q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = (half_t)out[c][4 * rq + e];
                        // columns 32 c + 8 rq + 4 lhi + 0..3 of token row l31
                        *(LDSP(half4_t))(own_row + (4 * c + rq) * 16) = q;
                    }
            }
            // the rows are read back as half8: a different vector type than the half4 stores above — without the fence
            // hipcc's type-based alias analysis lets the first row read overtake them
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const half8_t a = *(LDSP(const half8_t))(my_slot + i * TL_PITCH);
                const int soff = soff0 + (4 * ps + i) * (int)row_bytes;
                if (MODE & 4) {   // test decomposition: the raw GEMM result (no residual, no norm)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, a), rs_x, voff_row, soff, 0);
                    continue;
                }
                float v[8];
                float ss = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = (float)a[e] + (float)res[4 * ps + i][e] * p.alpha;
                    ss += v[e] * v[e];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                const float rstd = rsqrtf(ss / (float)TL_D + 1e-5f);
                half8_t y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (half_t)((v[e] * rstd) * w[e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y), rs_x, voff_row, soff, 0);
            }
            asm volatile("" ::: "memory");
        }
    };
    // this wave's 32 token rows as B fragments; fresh: rows this wave stored earlier in the launch (after a vmcnt(0))
    auto load_frags = [&](__amdgpu_buffer_rsrc_t rs, long row0, auto fresh) __attribute__((always_inline)) {
        const int soff0 = (int)(row0 * row_bytes);
        int vf = voff_frag;
        asm volatile("" : "+v"(vf));   // opaque per call: the 32 offsets below stay immediates instead of 32 hoisted registers
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
            xf[ks] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, vf + 32 * ks, soff0, decltype(fresh)::value ? 16 : 0));
    };

    // After every load_frags: the fragment ring must be back in VGPRs many instructions ahead of the asm MFMAs that read it.
    // Across the epilogue hipcc parks it in (then free) accumulator registers, and a v_accvgpr_read right in front of an asm
    // MFMA is the VALU -> MFMA-operand hazard tools/check_asm_hazards.py looks for.
#define TL_PIN_FR()                                                       \
    do {                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < TL_FD; ++i_) asm volatile("" : "+v"(fr[i_])); \
    } while (0)

    for (long ti = 0; ti < my_tiles; ++ti) {
        const long tile = blockIdx.x + ti * (long)gridDim.x;
        const long row0 = tile * 128 + wave * 32;       // first row of this wave
        // B fragments of the tile's input rows (attn, or x1 in the MLP-only mode)
        if (!(DBG & 8) || ti == 0) load_frags((MODE & 1) ? rs_attn : rs_x, row0, std::false_type{});
        TL_PIN_FR();
        if (MODE & 1) {
            // the out-proj accumulators start from the bias (column 32 c + 8 rq + 4 lhi + e of every token row)
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4_t bv = *(const float4_t *)(p.bo + 32 * c + 8 * rq + 4 * lhi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[c][4 * rq + e] = bv[e];
                }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[c][r] = 0.0f;
        }

        if (MODE & 1) {
            // ---- phase A: out-proj, 16 stages of 32 k ----
            cl_static_for<16>([&](auto st_c) __attribute__((always_inline)) {
                constexpr int st = decltype(st_c)::value;
                stage_512(xf[2 * st], xf[2 * st + 1]);
            });
            norm_rows(row0, p.n1, std::false_type{});     // p.x <- x1
            if (MODE == 1) continue;
            // x1 becomes the MLP's input fragments: re-read the rows this wave has just stored, once the stores are in L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_frags(rs_x, row0, std::true_type{});
            TL_PIN_FR();
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
                    else if (h == 1 && i == 30) tl_mfma_v_last(ay, a, xf[31]);
                    else if (h == 1 && i == 31) tl_mfma_v_last(ag, a, xf[31]);
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
                if (decltype(with_swiglu)::value && (i & 1) == 0) swiglu_elem(hn, i >> 1);
            });
        };
        // slab 0: FC1, SwiGLU alone
        fc1_slab();
#pragma unroll
        for (int r = 0; r < 16; ++r) swiglu_elem(ha, r);
        // slabs 1 .. NJ-1 in pairs (static fragment registers): odd slab -> hb, even slab -> ha
        for (int j = 1; j + 1 < NJ; j += 2) {
            fc1_slab();
            fc2_stage(ha, hb, std::true_type{});
            fc1_slab();
            fc2_stage(hb, ha, std::true_type{});
        }
        fc1_slab();                    // slab NJ-1 (odd): the last reader of the input fragments
        fc2_stage(ha, hb, std::true_type{});       // FC2 of slab NJ-2, SwiGLU of slab NJ-1
        fc2_stage(hb, ha, std::false_type{});      // FC2 of slab NJ-1

        // residual = x1 = the rows in p.x (stored by this wave in phase A, or the launch's input in the MLP-only mode)
        norm_rows(row0, p.n2, std::true_type{});
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
    const int dbg = mode >> 8;
    mode &= 0xff;
    const long ntiles = (R + 127) / 128;
    long grid = mibc_ncu();
    if (ntiles < grid) grid = ntiles;
#define TL_LAUNCH(M_)                                                                                   \
    do {                                                                                                \
        MIBC_LDS_ATTR_ONCE((tx_layer_kernel<M_>), TL_LDS_BYTES);                                        \
        hipLaunchKernelGGL((tx_layer_kernel<M_>), dim3((unsigned)grid), dim3(256), TL_LDS_BYTES, s, a); \
    } while (0)
#define TL_LAUNCH_DBG(D_)                                                                                  \
    do {                                                                                                   \
        MIBC_LDS_ATTR_ONCE((tx_layer_kernel<2, D_>), TL_LDS_BYTES);                                        \
        hipLaunchKernelGGL((tx_layer_kernel<2, D_>), dim3((unsigned)grid), dim3(256), TL_LDS_BYTES, s, a); \
        return 0;                                                                                          \
    } while (0)
    if (mode == 2 && dbg == 1) TL_LAUNCH_DBG(1);
    if (mode == 2 && dbg == 2) TL_LAUNCH_DBG(2);
    if (mode == 2 && dbg == 4) TL_LAUNCH_DBG(4);
    if (mode == 2 && dbg == 12) TL_LAUNCH_DBG(12);
    if (mode == 2 && dbg == 14) TL_LAUNCH_DBG(14);
    if (dbg != 0) return 1;
#undef TL_LAUNCH_DBG
    if (mode == 1) TL_LAUNCH(1);
    else if (mode == 2) TL_LAUNCH(2);
    else if (mode == 6) TL_LAUNCH(6);
    else TL_LAUNCH(3);
#undef TL_LAUNCH
    return 0;
}
