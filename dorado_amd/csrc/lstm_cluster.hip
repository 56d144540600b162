// dorado_amd/csrc/lstm_cluster.hip — hidden-split "CU cluster" LSTM layer for wide layers
// (C = 512 / 768 / 1024: sup@v4.3).  Same semantics as lstm.hip
// (torch LSTM, dorado/nn/LSTMStack.cpp:19-41; replaces host_cutlass_lstm, LSTMStack.cpp:193).
//
// Why: with one workgroup owning whole hidden vectors (lstm.hip, xg), a CU re-streams the layer's
// 4C x 2C weights (16.8 MB at C = 1024) for every 32 batch rows and every time step: 32 flop per
// weight byte, bound by the CU's L2 port at <= 0.42 of the MFMA peak.  Here KCL = C/128 workgroups
// (one per CU, 8 for C = 1024) form a CLUSTER that shares 256 batch rows; member j owns hidden units
// [128 j, 128 j + 128) for all T steps, i.e. a 512-column slice of the gate matrix, so its weight
// slice streams once per 256 rows (256 flop per weight byte).  Per step a member computes
//      gates[256 rows x 512] = [x_t | h_{t-1}] [256 x 2C] . W_j [2C x 512]
// in two passes of 64 hidden units (a 256 x 256 accumulator tile = 128 registers per lane), both
// operands staged HBM/L2 -> LDS by direct DMA (global_load_lds_dwordx4) through a 4-slot ring of
// K = 32 slabs (XOR-swizzled 64-byte rows; counted vmcnt, raw s_barrier: three slabs in flight),
// 8 waves = 4 row groups x 2 hidden groups on v_mfma_f32_16x16x32_f16 (round 6: was 32x32x16 — on random data the 32x32
// shapes pull the shader clock down to 1.2 GHz = 1.26 PF of bare MFMAs, the 16x16 shapes hold 1.87 GHz = 1.91 PF,
// tools/mfma_ref.hip; this kernel sat at 0.87 of the 32x32 ceiling).  The cell state lives in a
// private scratch buffer (fp32); gates are lane-local exactly as in lstm.hip.
//
// The exchange of h between the members IS the layer output: member j writes its 128-unit slice of
// h_t into Xout[t] with write-through (sc1) 16-byte stores, and every member reads the full h_{t-1}
// rows back as the second half of the next step's K range with sc1 DMA loads.  Hand-off protocol =
// cdna_hip_programming.md §6 Guideline 16, form R1 (placement-independent: correct for any
// workgroup -> XCD mapping; the same-XCD mapping below is a speed choice only):
//   producer  sc1 stores -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane
//             stores flag[cluster][j] = number of completed steps (relaxed, agent scope);
//   consumer  the flags are fetched (sc1) a few slabs before the first h slab is needed — all x-part
//             slabs of the step (no dependence on the exchange) come first, so the hand-off latency
//             hides behind ~13 us of MFMA work; a bounded blocking poll is the slow path.
// Every spin is bounded: on a time-out the workgroup records an error word, stops waiting for the
// rest of the launch (results are garbage, the kernel still terminates) and the host reports it.
//
// Round 4 — Q8: the reference's quantised path (nn/LSTMStack.cpp:127-211, KOI_I8; per-row weight scales
// utils::quantize_tensor :165-172) on the same machine mapping.  int8 activations (round(127 h), rows of C BYTES) and
// int8 weights keep the 64-byte slab rows, so every LDS image, DMA piece and fragment read is byte-identical to the f16
// kernel's; a slab now covers K = 64 (v_mfma_i32_16x16x64_i8: the lane's 16-byte fragment is 16 k-values), i.e. HALF the
// slabs per time step at the same MFMA count per slab.  Accumulators are int32 (exact), initialised with
// round(bias / deq[row]); gate pre-activation = float(acc) * deq[row], deq = 1 / (127 * row scale); gates, cell state and
// the h quantisation are fp32 as in the f16 kernel.  Layers 1 .. L-2 write int8 h (which is also the exchange); the last
// layer writes f16 for the CRF head AND an int8 copy of h (Hx) that its members exchange.  Opt-in
// (mibc_model_desc::lstm_quant), own tolerance (tests/test_gpu_baseline_parity.py).
#include "common.h"
#include "cluster_util.h"

typedef int int4q_t __attribute__((ext_vector_type(4)));
typedef int int16q_t __attribute__((ext_vector_type(16)));

#define CL_ROWS 256
#define CL_BK 32
#define CL_NST 4
#define CL_WTILE (256 * CL_BK)            // halfs per weight slab (16 KiB)
#define CL_ATILE (CL_ROWS * CL_BK)        // halfs per activation slab (16 KiB)
#define CL_STAGE (CL_WTILE + CL_ATILE)    // halfs per stage (32 KiB)
#define CL_PATCH_LD 40                    // halfs per patch row (32 + 8 pad)
#define CL_PATCH (32 * CL_PATCH_LD)       // halfs per wave patch
#define CL_SPIN_LIMIT 400000u

#define CL_OFF_PATCH (CL_NST * CL_STAGE * 2)                 // bytes
#define CL_OFF_BIAS (CL_OFF_PATCH + 8 * CL_PATCH * 2)
#define CL_OFF_DEQ (CL_OFF_BIAS + 2 * 2 * 4 * 32 * 4)        // Q8: dequantisation factors, bias order
#define CL_QPATCH_LD 48                                     // Q8: int8 h rows of a wave (32 batch rows x 32 hidden bytes) reuse its f16 patch
#define CL_OFF_FLAGZ (CL_OFF_DEQ + 2 * 2 * 4 * 32 * 4)
#define CL_OFF_SYNC (CL_OFF_FLAGZ + 256)
#define CL_LDS_BYTES (CL_OFF_SYNC + 64)
#ifndef MIBC_CL_WROW
#define MIBC_CL_WROW 0
#endif
// byte distance between consecutive K slabs of a weight slice: 64 B along a row (row-major image) or one 16 KiB slab image
#define CL_WSLAB (MIBC_CL_WROW ? 64u : (unsigned)(CL_WTILE * 2))

// A operand = 16 gate rows x K, B operand = 16 batch rows x K of one slab (K = 32 halfs / 64 int8: the lane's 16 bytes are the
// k-group lane >> 4 of row lane & 15); D[row = 4 (lane >> 4) + r][col = lane & 15]
__device__ __forceinline__ int4q_t cl_mfma_i8(half8_t a, half8_t b, int4q_t c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(int4q_t, a), __builtin_bit_cast(int4q_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ float4_t cl_mfma_f16(half8_t a, half8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// One launch = one layer.  grid = KCL * (clusters resident at once); a workgroup loops over the row
// groups (clusters of 256 rows) rg = first, first + stride, ... so that N may exceed one residency.
// DBG (debug build only; results are wrong when non-zero): timing ablations — 1 no gate math, 2 no DMA after the
// first step, 4 no MFMA, 8 no hand-off wait / publish, 16 no fragment reads, 32 / 64 weight / activation slabs
// always fetched from the same (cache-resident) address.
// Q8: 0 = f16; 1 = int8 in, int8 out; 2 = int8 in, f16 out + int8 exchange copy in Hx (last layer).
template <int C, bool MASKED, int DBG = 0, int Q8 = 0>
__global__ __launch_bounds__(512, 2) void lstm_layer_cl_kernel(
        const half_t *__restrict__ Xin,     // [T][N][C]  (Q8: int8 rows of C bytes)
        half_t *__restrict__ Xout,          // [T][N][C]  (Q8 == 1: int8)
        const half_t *__restrict__ Wt,      // [KCL][2][KS][256][32]: swizzled LDS images of the weight slabs (Q8: [256][64] int8)
        const float *__restrict__ biascl,   // [KCL][2][2][4][32]  (b_ih + b_hh); Q8: int32 round(bias / deq)
        const half_t *__restrict__ zeros,   // [256][C] zeros (h_{-1})
        float *__restrict__ cbuf,           // [N][C] f32 cell state, private layout (see seg_gates); no init needed
        unsigned *__restrict__ flags,       // [nclusters][KCL][16]: completed steps of member j (zeroed per launch)
        unsigned *__restrict__ err,         // [4]: sticky error word, first failing (cluster, step)
        int T, int N, int reverse, int cpx /* clusters per XCD slot group, 0 = linear map */,
        int resident_clusters,
        const unsigned long long *__restrict__ tmask /* MASKED: [T][N/64] */,
        const float *__restrict__ deqcl /* Q8: [KCL][2][2][4][32] */, signed char *__restrict__ Hx /* Q8 == 2: [T][N][C] int8 h */) {
    constexpr int KCL = C / 128;
    constexpr int EB = Q8 ? 1 : 2;         // bytes per activation / weight element
    constexpr int KSX = C * EB / 64;       // x-part slabs per pass (a slab row is 64 bytes: 32 halfs or 64 int8)
    constexpr int KS = 2 * KSX;            // slabs per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // every LDS access goes through explicit LDS-address-space pointers (no generic pointers, no aperture tests)
    LDSP(unsigned char) smem3 = (LDSP(unsigned char))smem;
    LDSP(half_t) stage = (LDSP(half_t))smem3;
    LDSP(half_t) patch_all = (LDSP(half_t))(smem3 + CL_OFF_PATCH);
    LDSP(float) bias_s = (LDSP(float))(smem3 + CL_OFF_BIAS);
    LDSP(float) deq_s = (LDSP(float))(smem3 + CL_OFF_DEQ);
    LDSP(unsigned) flagz = (LDSP(unsigned))(smem3 + CL_OFF_FLAGZ);
    LDSP(volatile unsigned) syncw = (LDSP(volatile unsigned))(smem3 + CL_OFF_SYNC);
    const unsigned lds0 = (unsigned)(size_t)smem3;   // byte address of the allocation (DMA destinations)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int rgw = wave >> 1, hg = wave & 1;

    // cluster / member of this workgroup.  Observed placement: block b runs on XCD b % 8 — putting the KCL
    // members of a cluster on one XCD makes the h exchange and the shared x_t rows L2-local (speed only).
    int cl0, j;
    {
        const int b = blockIdx.x;
        if (cpx > 0) {
            const int xcd = b & 7, slot = b >> 3;
            cl0 = xcd * cpx + slot / KCL;
            j = slot % KCL;
        } else {
            cl0 = b / KCL;
            j = b % KCL;
        }
    }
    const int nclusters = N / CL_ROWS;

    // DMA assignment: instruction q of this wave fills 16-byte slots [(wave*2+q)*64, +64) of a slab:
    // row = (wave*2+q)*16 + lane/4, physical 16-byte column lane%4 <- logical column (lane%4) ^ ((-(row>>2))&3)
    unsigned aoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        const int col = (lane & 3) ^ ((0 - (row >> 2)) & 3);
        aoff[q] = (unsigned)(row * C * EB + col * 16);       // bytes
    }

    for (int i = tid; i < 2 * 2 * 4 * 32; i += 512) {
        bias_s[i] = biascl[(size_t)j * 2 * 2 * 4 * 32 + i];
        // Q8: the dequantisation factor is stored pre-multiplied by -log2e (gates i, f, o) / -2 log2e (gate g; order [(p, hg)][gate][32]):
        // float(acc) * deq is then the exp2 argument of the gate's sigmoid / tanh directly (as in lstm_q8.hip, round 6)
        if (Q8) deq_s[i] = deqcl[(size_t)j * 2 * 2 * 4 * 32 + i] * ((((i >> 5) & 3) == 2) ? -2.88539008f : -1.44269504f);
    }
    LDSP(half_t) patch = patch_all + wave * CL_PATCH;
    LDSP(unsigned char) qpatch = (LDSP(unsigned char))patch;   // same memory: the f16 rows (if any) have left before the int8 rows are written
    bool dead = false;

    const bool grpB = wave >= 4;
    // XOR swizzle of the 16-byte column: (-(row >> 2)) & 3 — the four 16-lane groups a ds_read_b128 is serviced in
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md, LDS) then touch 16 distinct bank quads each under the
    // 16x16x32 fragment mapping (lane = row & 15 | k-group << 4); every fragment row of this lane has the same (row >> 2) & 3
    const int sw = (0 - (l15 >> 2)) & 3;
    const int cq = (lq ^ sw) << 3;              // physical 16-byte column of this lane's k-group (halfs)
    // fragment read bases of this lane inside a stage (halfs): gate-row block (g, hb) at + (g * 32 + hb * 16) rows, batch block bb at + 16 bb rows
    const int woff = (hg * 128 + l15) * CL_BK + cq, xoff = CL_WTILE + (rgw * 64 + l15) * CL_BK + cq;
    const unsigned long long wslice = (unsigned long long)(Wt + (size_t)j * 2 * KS * CL_WTILE);   // uniform
    const unsigned wlane = (unsigned)(((wave * 2) * 512 + lane * 8) * 2);                              // bytes
    // MIBC_CL_WROW = 1 (experiment, round 4; measured and NOT adopted: 207 vs 200 ms per f16 layer, 114 vs 112 int8): the
    // member's weights stored ROW-major ([pass][256 gate rows][2C], 64-byte slab segments 2C * EB bytes apart), a DMA piece
    // gathering 16 rows x 64 B exactly like an activation piece instead of reading 1 KiB of a pre-swizzled slab image
    const unsigned woffb[2] = {aoff[0] * 2u - (aoff[0] & 63u), aoff[1] * 2u - (aoff[1] & 63u)};     // row pitch 2C * EB, same 16-byte column
    const unsigned aoffb[2] = {aoff[0], aoff[1]};                                                       // bytes
    const unsigned dma_lds = lds0 + (unsigned)(wave * 2) * 1024u;   // + slot * 64 KiB/2 ... + tile + q * 1 KiB

    for (int cl = cl0; cl < nclusters; cl += resident_clusters) {
        const int n0 = cl * CL_ROWS;
        unsigned *myflag = flags + ((size_t)cl * KCL + j) * 16;
        const unsigned *clflags = flags + (size_t)cl * KCL * 16;

        // Per pass the 2C/32 slabs are FULLY UNROLLED: ring slots, weight-slab offsets and activation-column
        // offsets are compile-time constants relative to a handful of per-pass scalars, so a slab costs ~a dozen
        // scalar instructions (eight waves share the CU's scalar issue: a per-slab address state machine of ~150
        // scalar instructions alone took as long as the slab's MFMAs).
        // (Addresses are integers: selects between pointers into different allocations trip an address-space
        // inference bug of this hipcc: "Illegal instruction ... V_CMP_NE_U32 0, $src_shared_base".)
        const long long dstep = (reverse ? -1LL : 1LL) * (long long)N * C * EB;          // bytes per time step
        const int t_first = reverse ? (T - 1) : 0;
        unsigned long long x_cur = (unsigned long long)Xin + (unsigned long long)EB * (((size_t)t_first * N + n0) * C);
        unsigned long long h_cur = (unsigned long long)zeros;                              // h_{-1} = 0
        // the rows the members exchange: the layer output itself, or (Q8 == 2: f16 output) its int8 copy
        const unsigned long long o_first = (Q8 == 2 ? (unsigned long long)Hx : (unsigned long long)Xout) +
                                           (unsigned long long)EB * (((size_t)t_first * N + n0) * C);

        // DMA of one slab into ring slot `slot`: weights from byte offset w_off of this member's slice,
        // activations from a_base (row 0 of the cluster, first column of the slab).  Both activation halves
        // (x_t and the exchanged h_{t-1}) use agent-scope (sc1) loads: no L1 reuse to lose, and the hand-off needs it.
        // Every address is (uniform 64-bit base in SGPRs) + (this lane's constant 32-bit byte offset): the bases are
        // made opaque per slab — with 64 unrolled slabs the optimiser otherwise hoists 64 x 4 loop-invariant per-lane
        // 64-bit addresses out of the time loop and spills them.
        auto issue = [&](int slot, unsigned w_off, unsigned long long a_base, bool on) __attribute__((always_inline)) {
            if ((DBG & 2) && !on) return;
            const unsigned l = dma_lds + (unsigned)slot * (CL_STAGE * 2);
            if (DBG & 32) w_off = 0;                       // ablation: weight slab always L2-resident
            if (DBG & 64) a_base = (unsigned long long)Xin;   // ablation: activation slab always L2-resident
            unsigned long long wb = wslice + w_off;
            asm volatile("" : "+s"(wb));
            asm volatile("" : "+s"(a_base));
            if (MIBC_CL_WROW) {
                cl_dma16((ghalf_p)(wb + woffb[0]), l);
                cl_dma16((ghalf_p)(wb + woffb[1]), l + 1024);
            } else {
                ghalf_p wp = (ghalf_p)(wb + wlane);
                cl_dma16(wp, l);
                cl_dma16(wp + 512, l + 1024);
            }
            cl_dma16_sc1((ghalf_p)(a_base + aoffb[0]), l + CL_WTILE * 2);
            cl_dma16_sc1((ghalf_p)(a_base + aoffb[1]), l + CL_WTILE * 2 + 1024);
        };

        __syncthreads();   // bias_s visible; previous row group's LDS reads are done
        issue(0, 0u, x_cur, true);                       // slabs 0 and 1 of (step 0, pass 0): 2 slabs ahead
        issue(1, CL_WSLAB, x_cur + 2ull * CL_BK, true);

        // ---- two wave groups in anti-phase ----
        // Waves w and w + 4 share a SIMD.  Group A = waves 0-3, group B = waves 4-7; per slab every wave runs a LOAD
        // segment L (issue the DMAs of slab g + 2, read the first k16 step's fragments of slab g) and a MATRIX
        // segment M (16 MFMAs, the second step's fragments fetched in a rolling fashion behind the MFMAs that free
        // their registers), with a workgroup barrier in front of each:  B1 L(g) B2 M(g)  B1 L(g+1) B2 M(g+1) ...
        // Both groups run the SAME instruction stream; group B executes one extra barrier up front (group A one at
        // the end), which shifts B by one barrier instance: A.B1(g) pairs with B.B2(g-1), A.B2(g) with B.B1(g) —
        // on every SIMD one wave feeds the matrix pipe while its partner occupies the LDS / DMA / scalar issue.
        // Slab g must have landed in LDS — every wave's part — before the instance after which the first wave reads
        // it, A.B1(g) == B.B2(g-1): with 4 DMAs per lane and slab and 2 slabs of look-ahead, A waits vmcnt(4) before
        // its B1(g), B waits vmcnt(4) before its B2(g-1) (younger: one slab).  Extra younger operations (h stores,
        // flag fetch) only make a wait stricter.  The DMAs of slab g + 2 reuse the ring slot of slab g - 2, last
        // read in M(g-2), which both groups have finished before the instance A.B1(g).
        float4_t acc[4][2][4];     // [gate][hidden block of 16][batch block of 16]
        int4q_t acq[4][2][4];      // Q8 accumulators (the unused set is dead code)
        half8_t wf[4], xq[4];
        float4_t cpre[4][2];       // [batch block][hidden block]
        // ---- gates of one pass (gp, time index gt): the lane holds, for batch row bb * 16 + l15 and hidden units hb * 16 + 4 lq + e,
        // all four gate pre-activations.  Cell state: fp32, in a private scratch buffer instead of registers (the registers buy
        // the fragment double-buffering of the anti-phase schedule).  Layout [cluster][member][pass][wave][bb][hb][lane][4]: every
        // access is one coalesced 1 KiB wave transaction; 32 KiB per workgroup and pass.
        auto gates = [&](int gp, int gt, bool gfirst, unsigned long long gvm) __attribute__((always_inline)) {
            if (gfirst || (DBG & 128)) {   // zero initial state (nn/LSTMStack.cpp:29-41: no h0 / c0 given); DBG 128: ablation, no c loads
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) cpre[bb][hb] = (float4_t)(0.0f);
            } else {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
                        cpre[bb][hb] = *((const float4_t *)(cbuf + ((((((size_t)cl * KCL + j) * 2 + gp) * 8 + wave) * 4 + bb) * 2 + hb) * 256) + lane);
            }
            const int hcol = j * 128 + gp * 64 + hg * 32;
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(Xout + ((size_t)gt * N + n0 + rgw * 64) * C + hcol), 0, 64 * C * 2, 0x00020000);
            // Q8: the int8 rows (the layer output when Q8 == 1, the exchange copy Hx when Q8 == 2)
            const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)((Q8 == 2 ? Hx : (signed char *)Xout) + ((size_t)gt * N + n0 + rgw * 64) * C + hcol), 0, 64 * C, 0x00020000);
            // (1) the arithmetic of all 32 elements of the lane — 8 independent (batch block, hidden block) groups of 4, so that the
            //     scheduler has dependent chains to interleave (the matrix pipe idles while a wave is here: the chains' latency, not
            //     the instruction count, is what this costs) — results kept as f16 quads / int8 quads; (2) the rows leave through the
            //     wave's patch, 32 batch rows at a time.
            half4_t hvv[4][2];
            int pkq[4][2];
            // f16 instance — staged and software-pipelined by hand: the straight-line form leaves hipcc working on two elements at a time
            // (exp -> add -> rcp chains back to back).  Per block of 4 hidden units x 4 gates: 16 independent exp2 arguments, then
            // 16 v_exp_f32, 16 adds, 16 v_rcp_f32 (stage 1); the cell update and its own exp / rcp (stage 2) of block n is issued behind
            // stage 1 of block n + 1.  Same operations per element, same results; same-box A/B: f16 LSTM stack 947.4 -> 935.3 ms.
            float r1[2][4][4];                     // [buffer][gate][e]: sigmoid / (tanh + 1) / 2 of the four gates
            auto stage1 = [&](int blk, float (&r)[4][4]) __attribute__((always_inline)) {
                const int bb = blk >> 1, hb = blk & 1;
                float x[4][4];
                if (Q8) {
                    float4_t dq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) dq[g] = *(LDSP(const float4_t))(deq_s + ((gp * 2 + hg) * 4 + g) * 32 + hb * 16 + 4 * lq);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[g][e] = (float)acq[g][hb][bb][e] * dq[g][e];     // pre-scaled: the exp2 argument
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[g][e] = acc[g][hb][bb][e] * (g == 2 ? -2.88539008f : -1.44269504f);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[g][e] = __builtin_amdgcn_exp2f(x[g][e]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[g][e] = 1.0f + x[g][e];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[g][e] = __builtin_amdgcn_rcpf(x[g][e]);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto stage2 = [&](int blk, const float (&r)[4][4]) __attribute__((always_inline)) {
                const int bb = blk >> 1, hb = blk & 1;
                const bool rowon = !MASKED || ((gvm >> (bb * 16 + l15)) & 1ull);
                float4_t cn;
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gg = fmaf(2.0f, r[2][e], -1.0f);
                    cn[e] = fmaf(r[1][e], cpre[bb][hb][e], r[0][e] * gg);
                    t[e] = __builtin_amdgcn_exp2f(cn[e] * -2.88539008f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_rcpf(1.0f + t[e]);
                half4_t hv;
                int pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float hval = r[3][e] * fmaf(2.0f, t[e], -1.0f);
                    if (MASKED && !rowon) {
                        cn[e] = 0.0f;
                        hval = 0.0f;
                    }
                    hv[e] = (half_t)hval;
                    if (Q8) pk |= ((int)__builtin_rintf(hval * 127.0f) & 0xff) << (8 * e);
                }
                *((float4_t *)(cbuf + ((((((size_t)cl * KCL + j) * 2 + gp) * 8 + wave) * 4 + bb) * 2 + hb) * 256) + lane) = cn;
                hvv[bb][hb] = hv;
                pkq[bb][hb] = pk;
            };
            if (DBG & 1) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        half4_t hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            hv[e] = (half_t)(1e-3f * (Q8 ? (float)(acq[0][hb][bb][e] + acq[1][hb][bb][e] + acq[2][hb][bb][e] + acq[3][hb][bb][e])
                                                         : (acc[0][hb][bb][e] + acc[1][hb][bb][e] + acc[2][hb][bb][e] + acc[3][hb][bb][e])));
                        *((float4_t *)(cbuf + ((((((size_t)cl * KCL + j) * 2 + gp) * 8 + wave) * 4 + bb) * 2 + hb) * 256) + lane) = cpre[bb][hb];
                        hvv[bb][hb] = hv;
                        pkq[bb][hb] = 0;
                    }
            } else if (Q8) {
                // int8 instances: the straight-line form (the staged one spills 18 registers here and measured 1 % slower,
                // profiles/r06_s_cl_gates_staged_ab.log)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const bool rowon = !MASKED || ((gvm >> (bb * 16 + l15)) & 1ull);
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        half4_t hv;
                        float4_t cn;
                        float4_t dq[4];      // dequantisation factors of this lane's 4 hidden units, per gate (pre-scaled: exp2 arguments)
                        int pk = 0;          // round(127 h) of the 4 units, one byte each
#pragma unroll
                        for (int g = 0; g < 4; ++g) dq[g] = *(LDSP(const float4_t))(deq_s + ((gp * 2 + hg) * 4 + g) * 32 + hb * 16 + 4 * lq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((float)acq[0][hb][bb][e] * dq[0][e]));
                            const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((float)acq[1][hb][bb][e] * dq[1][e]));
                            const float gg = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((float)acq[2][hb][bb][e] * dq[2][e])), -1.0f);
                            const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((float)acq[3][hb][bb][e] * dq[3][e]));
                            float c = fmaf(fg, cpre[bb][hb][e], ig * gg);
                            float hval = og * fast_tanh(c);
                            if (MASKED && !rowon) {
                                c = 0.0f;
                                hval = 0.0f;
                            }
                            cn[e] = c;
                            hv[e] = (half_t)hval;
                            pk |= ((int)__builtin_rintf(hval * 127.0f) & 0xff) << (8 * e);
                        }
                        *((float4_t *)(cbuf + ((((((size_t)cl * KCL + j) * 2 + gp) * 8 + wave) * 4 + bb) * 2 + hb) * 256) + lane) = cn;
                        hvv[bb][hb] = hv;
                        pkq[bb][hb] = pk;
                    }
                }
            } else {
                stage1(0, r1[0]);
#pragma unroll
                for (int blk = 0; blk < 8; ++blk) {
                    if (blk + 1 < 8) stage1(blk + 1, r1[(blk + 1) & 1]);
                    stage2(blk, r1[blk & 1]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {          // 32 batch rows at a time through the wave's patch
                if (Q8 != 1) {
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb)
                            *(LDSP(half4_t))(patch + (b2 * 16 + l15) * CL_PATCH_LD + hb * 16 + 4 * lq) = hvv[rt * 2 + b2][hb];
                }
                __builtin_amdgcn_wave_barrier();   // same wave: LDS operations execute in order
                if (Q8 != 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int prow = (lane >> 2) + 16 * i, seg = lane & 3;
                        const half8_t v = *(LDSP(const half8_t))(patch + prow * CL_PATCH_LD + seg * 8);
                        __builtin_amdgcn_raw_buffer_store_b128(
                                __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), ors,
                                ((rt * 32 + prow) * C + seg * 8) * 2, 0, Q8 ? 0 : 16 /* f16 exchange: sc1 write-through */);
                    }
                }
                if (Q8) {   // int8 rows of 32 bytes: one 16-byte piece per lane (the exchange: write-through)
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb) *(LDSP(int))(qpatch + (b2 * 16 + l15) * CL_QPATCH_LD + hb * 16 + 4 * lq) = pkq[rt * 2 + b2][hb];
                    __builtin_amdgcn_wave_barrier();
                    const int prow = lane >> 1, seg = lane & 1;
                    const int4q_t v = *(LDSP(const int4q_t))(qpatch + prow * CL_QPATCH_LD + seg * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), qrs,
                            (rt * 32 + prow) * C + seg * 16, 0, 16 /* sc1 */);
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        // pass whose gates group A still owes (it runs them behind L(0) of the NEXT pass, i.e. beside group B's last
        // M + gates of that pass, which come half a slab later by construction, instead of before them)
        int g_p = 0, g_t = 0;
        bool g_first = true, g_pending = false;
        unsigned long long g_vm = ~0ull;
        if (grpB) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // own part of slab 0 (group A reads it after this instance)
            __builtin_amdgcn_s_barrier();
        }
#pragma nounroll
        for (int step = 0; step < T; ++step) {
            const int t = reverse ? (T - 1 - step) : step;
            const bool last_step = (step == T - 1);
            unsigned long long vm = ~0ull;
            if (MASKED) vm = tmask[(size_t)t * (N / 64) + (n0 >> 6) + rgw];
            const unsigned long long x_next = last_step ? x_cur : (unsigned long long)((long long)x_cur + dstep);
#pragma nounroll
            for (int p = 0; p < 2; ++p) {
                const unsigned w_pass = (unsigned)p * (KS * CL_WTILE * 2);                 // bytes
                // the pass after this one (for the two look-ahead slabs at the end): pass 1 of this step, or pass 0
                // of the next step; past the end of the layer the last slabs are fetched again (nobody reads them)
                const unsigned w_n = (p == 0) ? (unsigned)(KS * CL_WTILE * 2) : 0u;
                const unsigned long long x_n = (p == 0) ? x_cur : x_next;
                const bool ev = (p == 0) && (step > 0) && !(DBG & 8);
                const bool dma_on = (step == 0);   // DBG & 2: DMAs only during the first step

                // Hand-off events (workgroup-wide, attached to ONE barrier instance: A's B1 of the slab, B's B2 of the
                // slab before).  ks == 3: every wave drained its h stores ahead of this instance -> publish
                // h_{step-1}.  ks == KSX - 2: h_{step-1} of every member must be visible before the first h slab
                // (slab KSX) is fetched.
                auto events = [&](int ks) __attribute__((always_inline)) {
                    if (ks == 3) {
                        if (ev && tid == 0 && !dead)
                            __hip_atomic_store((gu32 *)myflag, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (ks == KSX - 2) {
                        if (ev && !dead) {
                            bool ok = true;
#pragma unroll
                            for (int m = 0; m < KCL; ++m) ok = ok && (flagz[m] >= (unsigned)step);
                            if (!ok) {   // slow path (workgroup-uniform: every wave read the same words)
                                if (wave == 0) {
                                    unsigned spins = 0;
                                    bool good;
                                    do {
                                        const unsigned v = __hip_atomic_load(
                                                (gu32 *)(clflags + (lane < KCL ? lane : 0) * 16), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
                                        good = __all(v >= (unsigned)step);
                                        if (!good) __builtin_amdgcn_s_sleep(16);
                                    } while (!good && ++spins < CL_SPIN_LIMIT);
                                    if (lane == 0) syncw[0] = good ? 1u : 2u;
                                }
                                __syncthreads();
                                if (syncw[0] == 2u) {
                                    dead = true;
                                    if (tid == 0) atomicCAS(err, 0u, 0x80000000u | ((unsigned)cl << 16) | (unsigned)step);
                                }
                                __syncthreads();
                            }
                        }
                    }
                };

                cl_static_for<KS>([&](auto ks_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value;
                    constexpr int slot = ks & 3;
                    // ---------------- B1(g) ----------------
                    if (!grpB) {
                        if (ks == 3 && ev) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (!grpB) events(ks);
                    // ---------------- L(g): fragments of the first k16 step, DMAs of slab g + 2 ----------------
                    {
                        if (wave == 0 && ks == KSX - 4 && ev)
                            cl_dma4_sc1(clflags + (lane < KCL ? lane : 0) * 16, lds0 + CL_OFF_FLAGZ);
                        LDSP(const half_t) sp = stage + slot * CL_STAGE;
                        if (DBG & 16) {
#pragma unroll
                            for (int a = 0; a < 4; ++a) wf[a] = (half8_t)((half_t)(0.001f * (a + 1)));
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) xq[bb] = (half8_t)((half_t)(0.002f + 0.001f * bb));
                        } else {
                            // gate-row blocks 0-3 (= gates 0, 1 x hidden blocks 0, 1) and the four batch blocks of the slab
#pragma unroll
                            for (int a = 0; a < 4; ++a) wf[a] = *(LDSP(const half8_t))(sp + woff + ((a >> 1) * 32 + (a & 1) * 16) * CL_BK);
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) xq[bb] = *(LDSP(const half8_t))(sp + xoff + bb * 16 * CL_BK);
                        }
                        constexpr int LA = 2;
                        const int kt = ks + LA;
                        if (kt < KSX) {
                            issue(kt & 3, w_pass + (unsigned)kt * CL_WSLAB, x_cur + (unsigned)kt * (CL_BK * 2), dma_on);
                        } else if (kt < KS) {
                            issue(kt & 3, w_pass + (unsigned)kt * CL_WSLAB, h_cur + (unsigned)(kt - KSX) * (CL_BK * 2), dma_on);
                        } else {
                            issue(kt & 3, w_n + (unsigned)(kt - KS) * CL_WSLAB, x_n + (unsigned)(kt - KS) * (CL_BK * 2),
                                  dma_on && p == 0);
                        }
                        if (ks == 0 && !grpB && g_pending) gates(g_p, g_t, g_first, g_vm);
                    }
                    // ---------------- B2(g) ----------------
                    if (grpB) {
                        if (ks + 1 == 3 && ev) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (grpB && ks + 1 < KS) events(ks + 1);
                    // ---------------- M(g) ----------------
                    {
                        if (ks == 0) {   // first slab of a pass: accumulators start from the bias
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
#pragma unroll
                                for (int hb = 0; hb < 2; ++hb) {
                                    LDSP(const float) bp = bias_s + ((p * 2 + hg) * 4 + g) * 32 + hb * 16 + 4 * lq;
                                    const float4_t v = *(LDSP(const float4_t))bp;
                                    // (Q8: read the int32 words AS ints — hipcc folded bit_cast<int>(v[e]) of the float vector to
                                    // element 0 for all four e: every hidden unit of a lane started from its first unit's bias)
                                    const int4q_t vi = *(LDSP(const int4q_t))((LDSP(const int))bp);
#pragma unroll
                                    for (int bb = 0; bb < 4; ++bb) {
                                        if (Q8) acq[g][hb][bb] = vi;
                                        else acc[g][hb][bb] = v;
                                    }
                                }
                            }
                        }
                        LDSP(const half_t) sp = stage + slot * CL_STAGE;
                        __builtin_amdgcn_s_setprio(1);
                        __builtin_amdgcn_sched_barrier(0);
                        // 32 MFMAs per slab: gate-row block a = 2 g + hb against the four batch blocks; blocks 4-7 are fetched
                        // in a rolling fashion behind the MFMAs that free their registers
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) {
                                if (Q8) acq[a >> 1][a & 1][bb] = cl_mfma_i8(wf[a], xq[bb], acq[a >> 1][a & 1][bb]);
                                else if (!(DBG & 4)) acc[a >> 1][a & 1][bb] = cl_mfma_f16(wf[a], xq[bb], acc[a >> 1][a & 1][bb]);
                                else asm volatile("" ::"v"(wf[a]), "v"(xq[bb]));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (!(DBG & 16)) wf[a] = *(LDSP(const half8_t))(sp + woff + ((2 + (a >> 1)) * 32 + (a & 1) * 16) * CL_BK);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) {
                                if (Q8) acq[2 + (a >> 1)][a & 1][bb] = cl_mfma_i8(wf[a], xq[bb], acq[2 + (a >> 1)][a & 1][bb]);
                                else if (!(DBG & 4)) acc[2 + (a >> 1)][a & 1][bb] = cl_mfma_f16(wf[a], xq[bb], acc[2 + (a >> 1)][a & 1][bb]);
                                else asm volatile("" ::"v"(wf[a]), "v"(xq[bb]));
                            }
                        }
                        __builtin_amdgcn_s_setprio(0);
                        // group B: gates right behind its last MFMAs of the pass
                        if (ks == KS - 1 && grpB) gates(p, t, step == 0, vm);
                    }
                });
                g_p = p;
                g_t = t;
                g_first = (step == 0);
                g_vm = vm;
                g_pending = true;
            }
            // next step: x advances, h_{step} is this step's output rows
            h_cur = (step == 0) ? o_first : (unsigned long long)((long long)h_cur + dstep);
            x_cur = x_next;
        }
        if (!grpB) {
            gates(g_p, g_t, g_first, g_vm);   // group A's last pass
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

extern "C" size_t mibc_lstm_cl_lds_bytes(void) { return CL_LDS_BYTES; }

// Returns 0 if launched, 1 if the shape is not covered (caller falls back to the per-workgroup kernels).
MibcClusterGate &mibc_cluster_gate() {
    static MibcClusterGate g;
    return g;
}

// q8: 0 = f16 layer; 1 = int8 in / int8 out; 2 = int8 in / f16 out + int8 exchange copy hx (last layer).  For q8 != 0 Xin / Wt
// point at int8 data, biascl holds int32 round(bias / deq), deqcl the dequantisation factors.
extern "C" int mibc_launch_lstm_layer_cl(hipStream_t s, int C, const half_t *Xin, half_t *Xout, const half_t *Wt,
                                         const float *biascl, const half_t *zeros, float *cbuf, unsigned *flags,
                                         unsigned *err, int T, int N, int reverse,
                                         const unsigned long long *tmask, int q8, const float *deqcl, signed char *hx) {
    if (Wt == nullptr || biascl == nullptr || zeros == nullptr || cbuf == nullptr || flags == nullptr || err == nullptr)
        return 1;
    if (q8 != 0 && (deqcl == nullptr || (q8 == 2 && hx == nullptr) || q8 < 0 || q8 > 2)) return 1;
    if ((C != 512 && C != 768 && C != 1024) || N < CL_ROWS || N % CL_ROWS != 0) return 1;
    const int KCL = C / 128;
    const int nclusters = N / CL_ROWS;
    const int ncu = mibc_ncu();   // of the launching thread's current device
    // every member of a cluster must be resident at the same time: one workgroup per CU (the LDS request
    // guarantees it), never more workgroups than CUs
    int resident = ncu / KCL;
    if (resident < 1) return 1;
    if (resident > nclusters) resident = nclusters;
    int cpx = 0;
    if (resident % 8 == 0 && (resident / 8) * KCL * 8 <= ncu) cpx = resident / 8;   // same-XCD clusters
    const dim3 grid(resident * KCL);
    MibcClusterLaunch gate(s);   // after the previous cluster kernel of this device, on whatever stream
    if (hipMemsetAsync(flags, 0, (size_t)nclusters * KCL * 16 * sizeof(unsigned), s) != hipSuccess) return 1;
#define CL_LAUNCH(CC, M_)                                                                                   \
    do {                                                                                                    \
        MIBC_LDS_ATTR_ONCE((lstm_layer_cl_kernel<CC, M_>), CL_LDS_BYTES);                                   \
        hipLaunchKernelGGL((lstm_layer_cl_kernel<CC, M_>), grid, dim3(512), CL_LDS_BYTES, s, Xin, Xout, Wt, \
                           biascl, zeros, cbuf, flags, err, T, N, reverse, cpx, resident, tmask, nullptr, nullptr); \
    } while (0)
#define CL_LAUNCH_Q8M(CC, Q_, M_)                                                                           \
    do {                                                                                                    \
        MIBC_LDS_ATTR_ONCE((lstm_layer_cl_kernel<CC, M_, 0, Q_>), CL_LDS_BYTES);                            \
        hipLaunchKernelGGL((lstm_layer_cl_kernel<CC, M_, 0, Q_>), grid, dim3(512), CL_LDS_BYTES, s, Xin, Xout, Wt, \
                           biascl, zeros, cbuf, flags, err, T, N, reverse, cpx, resident, tmask, deqcl, hx); \
    } while (0)
    // (tmask != nullptr: the masked int8 instances — the reference's default GPU mode, variable chunks over the quantised LSTM)
#define CL_LAUNCH_Q8(CC, Q_)                                  \
    do {                                                      \
        if (tmask != nullptr) CL_LAUNCH_Q8M(CC, Q_, true);    \
        else CL_LAUNCH_Q8M(CC, Q_, false);                    \
    } while (0)
#ifdef MIBC_DEBUG_KERNELS
    // (debug library: timing ablations of the int8 -> int8 instance, MIBC_CL_DBG as for the f16 one; results are wrong)
    if (q8 == 1 && C == 1024 && tmask == nullptr) {
        static const int dbgq = MIBC_ENV_INT("MIBC_CL_DBG", 0);
#define CL_DBGQ(D_)                                                                                         \
    case D_: {                                                                                              \
        (void)hipFuncSetAttribute((const void *)lstm_layer_cl_kernel<1024, false, D_, 1>,                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CL_LDS_BYTES);                \
        hipLaunchKernelGGL((lstm_layer_cl_kernel<1024, false, D_, 1>), grid, dim3(512), CL_LDS_BYTES, s, Xin, \
                           Xout, Wt, biascl, zeros, cbuf, flags, err, T, N, reverse, cpx, resident, tmask, deqcl, hx); \
        return 0;                                                                                           \
    }
        switch (dbgq) { CL_DBGQ(1) CL_DBGQ(8) CL_DBGQ(32) CL_DBGQ(64) CL_DBGQ(96) default: break; }
#undef CL_DBGQ
    }
#endif
    if (q8 == 1) {
        switch (C) {
            case 512: CL_LAUNCH_Q8(512, 1); return 0;
            case 768: CL_LAUNCH_Q8(768, 1); return 0;
            default: CL_LAUNCH_Q8(1024, 1); return 0;
        }
    }
    if (q8 == 2) {
        switch (C) {
            case 512: CL_LAUNCH_Q8(512, 2); return 0;
            case 768: CL_LAUNCH_Q8(768, 2); return 0;
            default: CL_LAUNCH_Q8(1024, 2); return 0;
        }
    }
#ifdef MIBC_DEBUG_KERNELS
    static const int dbg = MIBC_ENV_INT("MIBC_CL_DBG", 0);
    if (dbg && C == 1024 && tmask == nullptr) {
#define CL_DBG(D_)                                                                                          \
    case D_: {                                                                                              \
        (void)hipFuncSetAttribute((const void *)lstm_layer_cl_kernel<1024, false, D_>,                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CL_LDS_BYTES);                \
        hipLaunchKernelGGL((lstm_layer_cl_kernel<1024, false, D_>), grid, dim3(512), CL_LDS_BYTES, s, Xin,  \
                           Xout, Wt, biascl, zeros, cbuf, flags, err, T, N, reverse, cpx, resident, tmask, nullptr, nullptr); \
        return 0;                                                                                           \
    }
        switch (dbg) { CL_DBG(1) CL_DBG(2) CL_DBG(4) CL_DBG(8) CL_DBG(16) CL_DBG(18) CL_DBG(22) CL_DBG(27) CL_DBG(32) CL_DBG(64) CL_DBG(96) CL_DBG(128) default: break; }
#undef CL_DBG
    }
#endif
    if (tmask != nullptr) {
        switch (C) {
            case 512: CL_LAUNCH(512, true); return 0;
            case 768: CL_LAUNCH(768, true); return 0;
            default: CL_LAUNCH(1024, true); return 0;
        }
    }
    switch (C) {
        case 512: CL_LAUNCH(512, false); return 0;
        case 768: CL_LAUNCH(768, false); return 0;
        default: CL_LAUNCH(1024, false); return 0;
    }
#undef CL_LAUNCH
#undef CL_LAUNCH_Q8
#undef CL_LAUNCH_Q8M
}
