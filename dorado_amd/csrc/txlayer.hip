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
//     LDS DMA (buffer_load_dwordx4 ... lds: resource + scalar stage offset + one per-lane VGPR offset), in CONSUMPTION order
//     and in MFMA-fragment order (the host lays the image out: every ds_read_b128 is lane-linear, conflict-free, no
//     swizzle), two stages ahead, counted vmcnt + one raw barrier per stage (= 32 MFMAs per wave); every stage is read by
//     all four waves (LDS fragment traffic: 1 KB per MFMA and wave = 128 B/clk/CU at the matrix pipe's peak);
//   * ONE wave per SIMD issues in order, so everything that is not an MFMA has to fit into the ~28 issue cycles an MFMA
//     leaves: each MFMA slot carries one fragment read (TL_FD slots ahead of its use), at most one of the eight 1 KB DMA
//     pieces of a stage (a burst of eight stalls the wave 100-185 cycles per piece), and a third of a SwiGLU element.  A
//     sched_barrier per slot keeps hipcc from regrouping that (left alone it batched 8-12 LDS reads and put whole SwiGLU
//     slabs between two MFMAs: 68 cycles per MFMA over the MLP; slot-scheduled: 45);
//   * SwiGLU of slab j is cut into 48 parts that ride the 96 MFMA slots after the slab's FC1 (the FC2 stage of slab j - 1
//     and FC1 of slab j + 1); the FC1 outputs are first packed to f16 pairs so that FC1 of the next slab can reuse ay / ag;
//   * both residual RMSNorms run on the accumulators, in registers: a lane holds half of ONE token row (quads of columns
//     32 c + 8 q + 4 lhi + 0..3) and lane ^ 32 the other half, so a row's sum of squares is a register sum and one exchange;
//     the residual and the result move as 8-byte quads; x1 becomes the MLP's input fragments by one v_permlane32_swap per
//     dword (and the fragments turn back into residual quads the same way): x1 is stored for the next layer's residual but
//     never read back.  The out-proj adds its bias first instead of last, FC2 sums the same products in a different order
//     inside each 16-group, and the norm sums squares in a different order, so x differs from the unfused path by f32
//     rounding only (tests/test_gpu_txlayer.py: max 0.004, rms 2e-5 at |x| <= 2.2).
// Cycle stamps at 1 M tokens (tools/txlayer_time.py, DBG 64, workgroup 0 tile 1, 128 rows):
//      out-proj 40 k | norm 1 44 k | MLP 287 k (6144 MFMAs: 47 cycles each, 32 = matrix-pipe peak) | norm 2 31 k
// 7.16 ms per layer against 9.2-9.4 ms for the five launches.
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
#ifndef TL_NORM_STORE16
#define TL_NORM_STORE16 1            // 1 = the norms store 16-byte fragments, data registers held until the stores are done (see norm_rows);
                                     // round 4, same box A/B: 7.087 -> 6.979 ms per layer (norm 1 45.8 k -> 36.5 k cycles, norm 2 36.1 k -> 23.6 k)
#endif
#ifndef TL_SWAP_STORE_NOPS
#define TL_SWAP_STORE_NOPS 7
#endif
#define TL_STR2(x) #x
#define TL_STR(x) TL_STR2(x)
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
    unsigned long long *trace;   // test-only: cycle stamps of workgroup 0 / wave 0 on its second tile (nullptr in production)
    // DBG template parameter = timing ablations (wrong results): 1 = every request fetches stage 0 (always L2-hot), 2 = no
    // requests at all, 4 = every fourth fragment straight from L2 into VGPRs instead of through LDS; 64 = cycle stamps into trace
    // (results unchanged)
};

// MODE 3 = whole tail; 1 = out-proj + norm 1 only (x <- x1); 2 = MLP + norm 2 only (x holds x1); 6 = MLP only, x <- the raw
// FC2 result   [test decomposition]
typedef half_t tl_half2 __attribute__((ext_vector_type(2)));
typedef unsigned tl_u2 __attribute__((ext_vector_type(2)));
typedef unsigned tl_u4 __attribute__((ext_vector_type(4)));

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
    // buffer addressing: resource in SGPRs, one per-lane VGPR offset for the whole kernel, stage / piece offsets scalar —
    // a 64-bit per-lane address got spilled, and every reload in the loop came with an s_waitcnt vmcnt(0), i.e. a drain of
    // all requests in flight
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)p.wimg + (size_t)first_stage * TL_STAGE_BYTES), 0, 0x7ffffffe, 0x00020000);
    const int voff_w = wave * 8 * 1024 + lane * 16;
    const unsigned dma_dst = lds0 + (unsigned)(wave * 8 * 1024);
    int g_issue = 0;        // next stage to request (ring slot g_issue & 3)
    int st_issue = 0;       // g_issue % stages_per_tile: stage of the image (past the last tile the stream simply wraps:
                            // two stages nobody reads — the waits are counted, so the request count per stage is constant)
    // one 1 KB piece per call: a burst of eight stalls the issuing wave for 100-185 cycles per piece (the request queue
    // backs up), one piece behind an MFMA costs about 60, half of it in the MFMA's shadow
    auto issue_piece = [&](int q) __attribute__((always_inline)) {
        const int soff = ((DBG & 1) ? 0 : st_issue) * TL_STAGE_BYTES + q * 1024;
        const unsigned dst = dma_dst + (unsigned)(g_issue & 3) * TL_STAGE_BYTES + q * 1024;
        // DBG 4: the fragments of every fourth MFMA slot come straight from L2 (below) — their share of the stage is not staged
        if (!(DBG & 2) && !((DBG & 4) && (q & 3) == 3))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_vptr)(size_t)dst, 16, voff_w, soff, 0, 0);
        if (q == 7) {
            ++g_issue;
            st_issue = (st_issue + 1 == stages_per_tile) ? 0 : st_issue + 1;
        }
    };
    auto issue_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(q);
    };
    // ---- stage protocol.  Boundary B(s) (executed inside stage s - 1, TL_FD MFMAs before its end; B(0) up front):
    //   wait until my requests for stage s have landed (the 8 of stage s + 1 stay in flight), barrier (everybody's have;
    //   and everybody has issued — and consumed — every read of stage s - 2, whose slot the next request overwrites),
    //   request stage s + 2.  The fragment ring fr[] runs TL_FD MFMAs ahead of the matrix pipe, across stage borders.
    int g_use = 0;          // stage whose fragments are being requested from LDS next
    LDSP(const half8_t) fcur = (LDSP(const half8_t))smem3 + lane;
    LDSP(const half8_t) fnext = fcur;
    auto boundary = [&]() __attribute__((always_inline)) {
        // (DBG 4: 6 staged pieces per stage, and the 8 direct fragment loads issued since are younger than the stage waited for)
        if (DBG & 4) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fnext = (LDSP(const half8_t))(smem3 + (unsigned)(g_use & 3) * TL_STAGE_BYTES) + lane;
        ++g_use;
    };
    half8_t fr[TL_FD];
    // DBG 4 (timing ablation, results wrong; VERDICT r3 / DESIGN 6b item 4): tx_layer is bound by the LDS port — every weight
    // fragment is read by all four waves, 4 KB per MFMA slot + the 1 KB the DMA writes against the 4 KB (128 B/clk x 32) the LDS
    // delivers.  Here every fourth slot's fragment is loaded by each wave straight from L2 into VGPRs, TL_GD x 4 slots ahead, and
    // neither staged nor read from LDS: 3.75 KB of LDS traffic per slot, 1.75 KB through the CU's L2 port (2 KB per slot peak).
    constexpr int TL_GD = 4;
    tl_u4 gfr[TL_GD];
    int goff = 0;
    auto load_direct = [&]() __attribute__((always_inline)) -> tl_u4 {
        const tl_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, goff, 0);
        goff += 4096;
        if (goff >= stages_per_tile * TL_STAGE_BYTES) goff = 0;
        return v;
    };
    if constexpr ((DBG & 4) != 0) {
#pragma unroll
        for (int i = 0; i < TL_GD; ++i) gfr[i] = load_direct();
    }
    issue_stage();
    issue_stage();
    boundary();
#pragma unroll
    for (int q = 0; q < TL_FD; ++q) issue_piece(q);     // the others go out behind the first MFMAs of the first stage
    fcur = fnext;
#pragma unroll
    for (int i = 0; i < TL_FD; ++i) fr[i] = fcur[i * 64];
    // one stage = 32 fragments, consumed in order by 32 MFMAs: mf(i, fragment i)
    auto run_stage = [&](auto &&mf) __attribute__((always_inline)) {
        cl_static_for<32>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            constexpr bool direct = (DBG & 4) != 0 && (i & 3) == 3;
            half8_t a = fr[i % TL_FD];
            if constexpr (direct) a = __builtin_bit_cast(half8_t, gfr[(i >> 2) % TL_GD]);
            mf(i_c, a);                                              // everything below runs in this MFMA's shadow
            if (i == 32 - TL_FD) boundary();
            if (i >= 32 - TL_FD) issue_piece(i - (32 - TL_FD));      // the request for stage s + 2, one piece per MFMA slot:
            if (i < 8 - TL_FD) issue_piece(i + TL_FD);               // TL_FD pieces here, the rest in the next stage
            if constexpr (direct) gfr[(i >> 2) % TL_GD] = load_direct();   // for slot i + 4 TL_GD (slot i + TL_FD is direct too)
            else fr[i % TL_FD] = (i + TL_FD < 32) ? fcur[(i + TL_FD) * 64] : fnext[(i + TL_FD - 32) * 64];
            __builtin_amdgcn_sched_barrier(0);      // one fragment read per MFMA slot: hipcc otherwise batches 8-12 reads
        });
        fcur = fnext;
    };

    half8_t xf[32];          // this wave's 32 token rows x 512 as B fragments: xf[ks] = row l31, k = 16 ks + 8 lhi + 0..7
    float16_t out[16];       // 32 x 512 f32: out[c][r] = column 32 c + (r&3) + 8 (r>>2) + 4 lhi of token row l31
    LDSP(unsigned char) side = smem3 + TL_OFF_SIDE + wave * 8192;      // this wave's patch: the two norm weight vectors
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ((LDSP(float))side)[lane * 8 + e] = p.n1[lane * 8 + e];
        ((LDSP(float))side)[512 + lane * 8 + e] = p.n2[lane * 8 + e];
    }
    asm volatile("" ::: "memory");
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

    // residual RMSNorm of this wave's 32 rows, on the accumulators: a lane holds 256 of the 512 columns of ONE token row (l31) —
    // quads of columns 32 c + 8 q + 4 lhi + 0..3 — and lane ^ 32 holds the other 256, so the row sum is a register sum plus
    // one exchange; nothing goes through LDS.  (Rounds 1-2 of this kernel transposed the accumulators through an LDS patch to
    // reuse residual_rmsnorm_kernel's row-per-wave form: 45-50 k cycles per 32 rows, a quarter of the kernel.)
    //   v = f16(out) + alpha * res ;  y = f16((v * rsqrt(mean(v^2) + eps)) * w) -> p.x        (nn/TxModules.cpp:41-56, 478-486)
    // Residual: from p.x (8-byte quads), or from the input fragments xf[] when those ARE the residual rows (the MLP's input).
    // A fragment (k = 16 ks + 8 lhi + 0..7 of row l31) is the quad pair {q = 2 m, 2 m + 1} of column tile c (ks = 2 c + m)
    // after one v_permlane32_swap per dword: lanes l31 / l31 + 32 hold quads A0 B0 / A1 B1 and need fragments A0 A1 / B0 B1.
    // The swap is its own inverse, so the same two instructions turn fragments back into quads — and turn the y quads of the
    // first norm into the MLP's input fragments without a trip through memory.
    // Norm weights: f32 in this wave's 8 KB LDS patch (n1 at 0, n2 at 2048), read as broadcast float4s.
    const int voff_quad = l31 * (int)row_bytes + 8 * lhi;
    auto norm_rows = [&](long row0, int which, auto from_xf, auto to_xf) __attribute__((always_inline)) {
        const int soff0 = (int)(row0 * row_bytes);
        int vq = voff_quad;
        asm volatile("" : "+v"(vq));     // opaque per call: offsets stay immediates
        LDSP(const unsigned char) wrow = side + which * 2048 + 16 * (vq & 8 ? 1 : 0);
        constexpr bool RAW = (MODE & 4) != 0;     // test decomposition: the raw GEMM result (no residual, no norm)
        // residual rows from memory: in fragment form (16 bytes per lane, 32 contiguous bytes per row and instruction — the
        // 8-byte quad gathers cost 13 k cycles more per 32 rows), turned into quads by the same swap as the from_xf path
        constexpr bool FROM_MEM = !decltype(from_xf)::value && !RAW;
        tl_u4 rf[FROM_MEM ? 32 : 1];
        if (FROM_MEM) {
            int vf = voff_frag;
            asm volatile("" : "+v"(vf));
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) rf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vf + 32 * ks, soff0, 0);
        }
        float ss = 0.0f;
        if (!RAW) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    tl_u2 ra, rb;
                    {
                        const tl_u4 f = FROM_MEM ? rf[FROM_MEM ? 2 * c + m : 0] : __builtin_bit_cast(tl_u4, xf[2 * c + m]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(f[0], f[2], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(f[1], f[3], false, false);
                        ra[0] = s0[0]; ra[1] = s1[0]; rb[0] = s0[1]; rb[1] = s1[1];
                    }
                    const half4_t ha4 = __builtin_bit_cast(half4_t, ra), hb4 = __builtin_bit_cast(half4_t, rb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float va = (float)(half_t)out[c][8 * m + e] + (float)ha4[e] * p.alpha;
                        const float vb = (float)(half_t)out[c][8 * m + 4 + e] + (float)hb4[e] * p.alpha;
                        ss += va * va;
                        ss += vb * vb;
                        out[c][8 * m + e] = va;          // v replaces the accumulator (same register: read, then written)
                        out[c][8 * m + 4 + e] = vb;
                    }
                }
            ss += __shfl_xor(ss, 32, 64);
        }
        const float rstd = rsqrtf(ss / (float)TL_D + 1e-5f);
#if TL_NORM_STORE16
        int voff_frag_o = voff_frag;
        asm volatile("" : "+v"(voff_frag_o));
        tl_u4 hold[32];
#endif
        // 8-byte stores: with 16-byte (fragment-form) stores the results were wrong in rows 12-15 / 28-31 of every tile — the
        // swap below writes BOTH its operands, and a register that still is the data of a > 8-byte store in flight must not
        // be written for a wait state hipcc does not insert for the swap's source operand
        float4_t wa_n = {0, 0, 0, 0}, wb_n = {0, 0, 0, 0};
        if (!RAW) {
            wa_n = *(LDSP(const float4_t))(wrow);
            wb_n = *(LDSP(const float4_t))(wrow + 32);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                half4_t ya, yb;
                if (RAW) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ya[e] = (half_t)out[c][8 * m + e];
                        yb[e] = (half_t)out[c][8 * m + 4 + e];
                    }
                } else {
                    const float4_t wa = wa_n, wb = wb_n;
                    if (2 * c + m + 1 < 32) {      // the next pair's norm weights: one pair ahead of their use
                        wa_n = *(LDSP(const float4_t))(wrow + 64 * (2 * c + m + 1));
                        wb_n = *(LDSP(const float4_t))(wrow + 64 * (2 * c + m + 1) + 32);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ya[e] = (half_t)((out[c][8 * m + e] * rstd) * wa[e]);
                        yb[e] = (half_t)((out[c][8 * m + 4 + e] * rstd) * wb[e]);
                    }
                }
                const tl_u2 ua = __builtin_bit_cast(tl_u2, ya), ub = __builtin_bit_cast(tl_u2, yb);
#if TL_NORM_STORE16
                {
                    unsigned a0 = ua[0], a1 = ua[1], b0 = ub[0], b1 = ub[1];
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    unsigned f0 = s0[0], f1 = s1[0], f2 = s0[1], f3 = s1[1];
                    tl_u4 f;
                    f[0] = f0; f[1] = f1; f[2] = f2; f[3] = f3;
                    __builtin_amdgcn_raw_buffer_store_b128(f, rs_x, voff_frag_o + 32 * (2 * c + m), soff0, 0);
                    if (decltype(to_xf)::value) xf[2 * c + m] = __builtin_bit_cast(half8_t, f);
                    hold[2 * c + m] = f;     // see below: nothing may overwrite a store's data registers before the store is done
                }
#else
                __builtin_amdgcn_raw_buffer_store_b64(ua, rs_x, vq + 64 * c + 32 * m, soff0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(ub, rs_x, vq + 64 * c + 32 * m + 16, soff0, 0);
                if (decltype(to_xf)::value) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
                    tl_u4 f;
                    f[0] = s0[0]; f[1] = s1[0]; f[2] = s0[1]; f[3] = s1[1];
                    xf[2 * c + m] = __builtin_bit_cast(half8_t, f);
                }
#endif
            }
#if TL_NORM_STORE16
        // Correct by construction: every store's data registers stay allocated (hold[], or xf[] when the fragments are the
        // MLP's input) until all stores of the epilogue have completed — v_permlane32_swap writes both its operands and
        // nothing holds such a write back while a > 8-byte store still has to read its data.
        if (!decltype(to_xf)::value) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 32; k += 4) asm volatile("" :: "v"(hold[k]), "v"(hold[k + 1]), "v"(hold[k + 2]), "v"(hold[k + 3]));
        }
#endif
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
    // After an epilogue the ring is read again (the stage has long landed): the copies requested at the end of the stage
    // before it are then dead instead of living — spilled, and reloaded between the stores — through the epilogue.
#define TL_REPRIME_FR()                                                   \
    do {                                                                  \
        asm volatile("" ::: "memory");                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < TL_FD; ++i_) fr[i_] = fcur[i_ * 64]; \
        TL_PIN_FR();                                                      \
    } while (0)

#define TL_STAMP(k_)                                                                        \
    do {                                                                                    \
        if ((DBG & 64) && p.trace != nullptr && blockIdx.x == 0 && tid == 0 && ti == 1) p.trace[k_] = __builtin_readcyclecounter(); \
    } while (0)
    for (long ti = 0; ti < my_tiles; ++ti) {
        const long tile = blockIdx.x + ti * (long)gridDim.x;
        TL_STAMP(0);
        const long row0 = tile * 128 + wave * 32;       // first row of this wave
        // B fragments of the tile's input rows (attn, or x1 in the MLP-only mode)
        load_frags((MODE & 1) ? rs_attn : rs_x, row0, std::false_type{});
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
            TL_STAMP(1);
            // p.x <- x1, and x1 becomes the MLP's input fragments in place
            if (MODE == 1) {
                norm_rows(row0, 0, std::false_type{}, std::false_type{});
                TL_REPRIME_FR();
                continue;
            }
            norm_rows(row0, 0, std::false_type{}, std::true_type{});
            TL_STAMP(2);
            TL_REPRIME_FR();
            TL_STAMP(3);
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[c][r] = 0.0f;
        }

        // ---- phase B: gated MLP, slabs of 32 hidden units ----
        float16_t ay, ag;
        half8_t hh[2];
        // SwiGLU of a slab (nn/TxModules.cpp:171-175 on the f16-rounded FC1 outputs) -> FC2 B fragments.  One wave per SIMD
        // issues in order, and an MFMA's shadow holds about 28 cycles of other work; a SwiGLU element costs about 60 (v_exp and
        // v_rcp are quarter rate).  The sixteen elements of a slab are therefore cut into 48 parts and spread over the 96 MFMA
        // slots that follow the slab's FC1: its outputs are first packed to f16 pairs (which frees ay / ag for the next slab's
        // FC1), then part A = exp, B = reciprocal, C = products + store into the fragment.  (All sixteen between the MFMAs of
        // the one FC2 stage: 64 cycles per MFMA there.)
        tl_half2 yg[16];
        float sw_t = 0.0f;
        auto sw_pack = [&](int r) __attribute__((always_inline)) {
            tl_half2 v;
            v[0] = (half_t)ay[r];
            v[1] = (half_t)ag[r];
            asm volatile("" : "+v"(v));
            yg[r] = v;
        };
        auto sw_part = [&](half8_t (&hn)[2], int n) __attribute__((always_inline)) {
            const int e = n / 3, ph = n % 3;
            if (ph == 0) {
                sw_t = __expf(-(float)yg[e][1]);
            } else if (ph == 1) {
                sw_t = __builtin_amdgcn_rcpf(1.0f + sw_t);
            } else {
                const float gt = (float)yg[e][1], y = (float)yg[e][0];
                sw_t = gt * sw_t * y;
            }
            asm volatile("" : "+v"(sw_t));      // computed in this MFMA slot (hipcc otherwise sinks the lot to the first use)
            if (ph == 2) hn[e >> 3][e & 7] = (half_t)sw_t;
        };
        // FC1 of one slab: two stages of 256 k each; parts 16 .. 47 of the previous slab's SwiGLU behind every other MFMA
        auto fc1_slab = [&](half8_t (&hn)[2], auto with_swiglu) __attribute__((always_inline)) {
            cl_static_for<2>([&](auto h_c) __attribute__((always_inline)) {
                constexpr int h = decltype(h_c)::value;
                run_stage([&](auto i_c, half8_t a) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_c)::value;      // fragment (s = i / 2, t = i % 2), k-step ks = 16 h + s
                    if (h == 0 && i < 2) {
                        if ((i & 1) == 0) tl_mfma_v_first(ay, a, xf[0]);
                        else tl_mfma_v_first(ag, a, xf[0]);
                    } else if (h == 1 && i >= 30) {
                        if ((i & 1) == 0) tl_mfma_v_last(ay, a, xf[31]);
                        else tl_mfma_v_last(ag, a, xf[31]);
                    } else {
                        if ((i & 1) == 0) tl_mfma_v(ay, a, xf[16 * h + (i >> 1)]);
                        else tl_mfma_v(ag, a, xf[16 * h + (i >> 1)]);
                    }
                    if (decltype(with_swiglu)::value && (i & 1) == 0) sw_part(hn, 16 + 16 * h + (i >> 1));
                });
            });
        };
        // FC2 stage of slab j - 1 (fragments hp); with_swiglu: the FC1 outputs in ay / ag (slab j) are packed behind the first
        // sixteen MFMAs, parts 0 .. 15 of their SwiGLU (-> hn) behind the other sixteen
        auto fc2_stage = [&](const half8_t (&hp)[2], half8_t (&hn)[2], auto with_swiglu) __attribute__((always_inline)) {
            run_stage([&](auto i_c, half8_t a) __attribute__((always_inline)) {
                constexpr int i = decltype(i_c)::value;
                out[i & 15] = mfma32x32x16(a, hp[i >> 4], out[i & 15]);
                if (decltype(with_swiglu)::value) {
                    if (i < 16) sw_pack(i);
                    else sw_part(hn, i - 16);
                }
            });
        };
        // One fragment buffer hh is enough: the FC2 stage of slab j - 1 reads hh[0] behind its first sixteen MFMAs and hh[1]
        // behind the others, and the first SwiGLU parts of slab j that complete there (elements 0-4, all in hh[0]) start at
        // MFMA sixteen; the remaining elements complete during FC1 of slab j + 1, after the stage.
        // slab 0: FC1, SwiGLU alone
        fc1_slab(hh, std::false_type{});
#pragma unroll
        for (int r = 0; r < 16; ++r) sw_pack(r);
#pragma unroll
        for (int n = 0; n < 48; ++n) sw_part(hh, n);
        fc1_slab(hh, std::false_type{});
        // invariant: ay / ag = FC1 of slab j, hh = SwiGLU of slab j - 1
        for (int j = 1; j + 1 < NJ; ++j) {
            if (j == 9) TL_STAMP(8);
            fc2_stage(hh, hh, std::true_type{});      // FC2 of slab j - 1
            if (j == 9) TL_STAMP(9);
            fc1_slab(hh, std::true_type{});           // FC1 of slab j + 1 (the last one = the last reader of the input fragments)
            if (j == 9) TL_STAMP(10);
        }
        fc2_stage(hh, hh, std::true_type{});          // FC2 of slab NJ - 2, first parts of the SwiGLU of slab NJ - 1
#pragma unroll
        for (int n = 16; n < 48; ++n) sw_part(hh, n);
        fc2_stage(hh, hh, std::false_type{});         // FC2 of slab NJ - 1

        TL_STAMP(4);
        // residual = x1 = the MLP's input fragments
        norm_rows(row0, 1, std::true_type{}, std::false_type{});
        TL_REPRIME_FR();
        TL_STAMP(5);
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

#ifdef MIBC_DEBUG_KERNELS
// test hook (host only, debug library): the image as f16 bit patterns; returns the number of halfs (or that number alone if out == nullptr)
MIBC_HOOK long mibc_debug_tx_layer_image(const float *wo, const float *w1, const float *w2, int FF, uint16_t *out, long cap) {
    if (FF < 64 || FF % 64 != 0) return -1;
    const std::vector<half_t> img = tx_layer_image(wo, w1, w2, FF);
    if (out) {
        if ((long)img.size() > cap) return -2;
        memcpy(out, img.data(), img.size() * sizeof(half_t));
    }
    return (long)img.size();
}
#endif

bool tx_layer_supported(int d_model, int ff) { return d_model == TL_D && ff >= 128 && ff % 64 == 0; }

// mode: 3 whole layer tail, 1 out-proj + norm 1 only, 2 MLP + norm 2 only (tests).  0 = launched, 1 = shape not covered.
static unsigned long long *g_tl_trace = nullptr;   // test hook (mibc_debug_txlayer_trace, debug library only)
#ifdef MIBC_DEBUG_KERNELS
MIBC_HOOK void mibc_debug_txlayer_trace(unsigned long long *dev_buf) { g_tl_trace = dev_buf; }
#endif

extern "C" int mibc_launch_tx_layer(hipStream_t s, const half_t *attn, half_t *x, const half_t *wimg, const float *bo,
                                    const float *n1, const float *n2, float alpha, long R, int FF, int mode) {
    if (!tx_layer_supported(TL_D, FF) || R <= 0) return 1;
    // The kernel addresses rows through 32-bit buffer offsets (row * 1024 bytes): rows are independent, so a call with more
    // than 2^20 rows (sup@v5 batches over 1024 chunks) is issued as consecutive launches of at most 2^20 rows each.
    constexpr long RMAX = 1L << 20;
    if (R > RMAX) {
        for (long r0 = 0; r0 < R; r0 += RMAX) {
            const long rp = (R - r0 < RMAX) ? (R - r0) : RMAX;
            const int rc = mibc_launch_tx_layer(s, attn + r0 * TL_D, x + r0 * TL_D, wimg, bo, n1, n2, alpha, rp, FF, mode);
            if (rc != 0) return rc;
        }
        return 0;
    }
    TxLayerArgs a{attn, x, wimg, bo, n1, n2, alpha, R, FF, g_tl_trace};
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
#define TL_LAUNCH_DBG(M_, D_)                                                                              \
    do {                                                                                                   \
        MIBC_LDS_ATTR_ONCE((tx_layer_kernel<M_, D_>), TL_LDS_BYTES);                                        \
        hipLaunchKernelGGL((tx_layer_kernel<M_, D_>), dim3((unsigned)grid), dim3(256), TL_LDS_BYTES, s, a); \
        return 0;                                                                                          \
    } while (0)
#ifdef MIBC_DEBUG_KERNELS   // ablation instances and the partial test modes exist only in the debug library
    if (mode == 2 && dbg == 1) TL_LAUNCH_DBG(2, 1);
    if (mode == 2 && dbg == 2) TL_LAUNCH_DBG(2, 2);
    if (mode == 2 && dbg == 64) TL_LAUNCH_DBG(2, 64);
    if (mode == 2 && dbg == 66) TL_LAUNCH_DBG(2, 66);
    if (mode == 2 && dbg == 4) TL_LAUNCH_DBG(2, 4);
    if (mode == 2 && dbg == 68) TL_LAUNCH_DBG(2, 68);
    if (mode == 3 && dbg == 4) TL_LAUNCH_DBG(3, 4);
    if (mode == 3 && dbg == 64) TL_LAUNCH_DBG(3, 64);
    if (dbg != 0) return 1;
    if (mode == 1) { TL_LAUNCH(1); return 0; }
    if (mode == 2) { TL_LAUNCH(2); return 0; }
    if (mode == 6) { TL_LAUNCH(6); return 0; }
#endif
#undef TL_LAUNCH_DBG
    if (mode != 3 || dbg != 0) return 1;
    TL_LAUNCH(3);
#undef TL_LAUNCH
    return 0;
}
