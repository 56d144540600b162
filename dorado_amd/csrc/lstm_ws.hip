// dorado_amd/csrc/lstm_ws.hip — WEIGHT-STATIONARY CU-cluster LSTM layer for C = 384 (hac@v4.3).
// Same semantics as lstm.hip (torch LSTM, dorado/nn/LSTMStack.cpp:19-41; replaces host_cutlass_lstm,
// LSTMStack.cpp:193) and, element for element, the same arithmetic as lstm_layer_x8_kernel (same MFMA shape,
// same k order per accumulator, same gate functions): the two are bit-identical.
//
// Why: lstm_layer_x8_kernel re-streams the layer's 2.36 MB of weights from L2 on every one of T time steps in every
// CU (64 rows per CU): 64 B/clk/CU, exactly the CU's L2 port, which together with the LDS fragment reads and the
// power-limited clock holds it at 0.45 of the MFMA peak.  Here the weights never move: KCL = C/64 = 6 workgroups of
// one XCD (one per CU, 8 waves) form a cluster; member j owns hidden units [64 j, 64 j + 64); the two waves of SIMD
// s own the 16 units [64 j + 16 s, +16): the X-WAVE keeps the W_ih fragments of those units' four gates (12 k-steps
// x 4 gates x 4 registers = 192 registers), the H-WAVE the W_hh fragments — resident for the whole launch (128 in
// AGPRs, read by v_mfma_f32_16x16x32_f16 directly as srcA).
// A cluster owns R row tiles of 16 batch rows; per time step it walks them.  For tile i the x-wave computes
// A = bias + x_t W_ih^T (12 k-steps) and hands the 16 x 16 x 4 accumulators to its partner through LDS; the h-wave
// CONTINUES the same accumulators with h_{t-1} W_hh^T (so the summation order is exactly x8's), applies the gates,
// keeps the fp32 cell state in a private scratch tile and stores its 16 x 16 block of h_t; the x-wave is one tile
// ahead.  One wave of a SIMD feeds the matrix pipe while the other one issues DMAs / gate math.
// Activations [x_t | h_{t-1}] (16 rows x 768, 24 KB) and the tile's cell state (4 KB) arrive by direct LDS DMA
// (issued by the x-waves) through two 4-slot rings (the halves of a tile are consumed one iteration apart).  h_t goes into Xout[t] — the layer output IS the exchange
// buffer — and is read back by all six members R tiles (one time step) later, so the hand-off latency is hidden by
// construction; progress counters per (member, SIMD) make it safe (cdna_hip_programming.md §6 Guideline 16 R1):
// an h-wave publishes "tiles complete" only after a counted s_waitcnt has retired its h stores, an x-wave checks a
// DMA-fetched snapshot of the 24 counters before it requests rows of h_{t-1} (bounded DMA-refresh poll as the slow
// path).  No VGPR-returning global load exists inside the loop (beside LDS-DMA traffic hipcc would drain vmcnt(0)
// before the first ds_read of every iteration).
// Measured history in DESIGN.md §4 (one wave per SIMD holding all 384 weight registers was issue-bound: 75 ms; this
// variant 64.5 ms against 59.9 for lstm_layer_x8_kernel on the same box, so the engine does not select it by default).
#include "common.h"
#include "cluster_util.h"

#define WS_TR 16                  // batch rows per tile
#define WS_LA 3                   // look-ahead in iterations: x half of tile i + 1 + LA and h half of tile i + LA are requested in iteration i
#define WS_NSL (WS_LA + 1)        // slots of each ring
#define WS_SPIN_LIMIT 400000u
#define WS_RMIN 12                // fewest row tiles per cluster (progress is published / observed a few tiles late)

typedef float float4v_ws __attribute__((ext_vector_type(4)));

// resident weights as srcA: from AGPRs ("a") or VGPRs ("v").  Inline asm on purpose: with the builtin the register
// allocator moves the accumulators into AGPRs and evicts resident weights.  The hazard recogniser does not see asm
// MFMAs, so every MFMA block ends with explicit wait states before its accumulators are read.
__device__ __forceinline__ float4v_ws ws_mfma_a(half8_t wa, half8_t b, float4v_ws c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(wa), "v"(b));
    return c;
}
__device__ __forceinline__ float4v_ws ws_mfma_v(half8_t wv, half8_t b, float4v_ws c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(wv), "v"(b));
    return c;
}

#ifdef MIBC_DEBUG_KERNELS
// DBG & 128: cycle stamps (s_memtime) of the phases of 16 consecutive iterations, cluster 0 / member 0 / SIMD 0,
// x-wave stamps in [it][0][k], h-wave in [it][1][k]
__device__ unsigned long long ws_trace[16 * 2 * 8];
#define WS_STAMP(role, k)                                                                        \
    do {                                                                                         \
        if ((DBG & 128) && tracing && i >= 1000 && i < 1016)                                     \
            ws_trace[((i - 1000) * 2 + (role)) * 8 + (k)] = __builtin_readcyclecounter();        \
    } while (0)
extern "C" int mibc_debug_ws_trace(unsigned long long *host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ws_trace), sizeof(ws_trace)) == hipSuccess ? 0 : -1;
}
#else
#define WS_STAMP(role, k) do { } while (0)
#endif

template <int C>
struct WsLayout {
    static constexpr int KCL = C / 64;
    static constexpr int KSX = C / 32;                        // k-steps per half (x or h)
    static constexpr int KS = 2 * KSX;
    static constexpr int KA = KSX < 8 ? KSX : 8;              // k-steps of a wave whose weights live in AGPRs
    static constexpr int ACT = KS * 1024;                     // bytes of one activation tile
    static constexpr int STAGE = ACT + 4096;                  // + the tile's cell state (4 x 1 KiB)
    static constexpr int XS = KSX * 1024;                     // x ring slot: the x_t half of a tile's activations
    static constexpr int HS = KSX * 1024 + 4096;              // h ring slot: the h_{t-1} half + the tile's cell state
    static constexpr int OFF_XR = 0, OFF_HR = WS_NSL * XS;
    static constexpr int OFF_HAND = OFF_HR + WS_NSL * HS;     // [2 slots][4 SIMDs][4 gates][64 lanes] float4
    static constexpr int OFF_BIAS = OFF_HAND + 2 * 4 * 4096;  // [4 SIMDs][4 gates][16] f32
    static constexpr int OFF_FLAGZ = OFF_BIAS + 1024;         // [4][64] u32 snapshot of the cluster's counters
    static constexpr int BYTES = OFF_FLAGZ + 1024;
    static constexpr int NF = KCL * 4;                        // counters per cluster (member x SIMD)
    static constexpr int XOPS = KS / 4 + 5;                   // x-wave VMEM operations per iteration: publish, h and c
                                                              // store, flags, KS/4 blocks, c block
};

// DBG (debug build only; results are wrong when non-zero): 1 no gate math, 2 no DMA after the prologue, 4 no MFMA,
// 8 no hand-off check, 16 no fragment reads, 32 no global stores.
template <int C, int DBG = 0>
__global__ __launch_bounds__(512) void lstm_layer_ws_kernel(
        const half_t *__restrict__ Xin,     // [T][N][C]
        half_t *__restrict__ Xout,          // [T][N][C]
        const half_t *__restrict__ Wf16,    // [C/16][2C/32][4][64][8]: lstm_layer_x8_kernel's fragment order
        const float *__restrict__ biasn,    // [4C]: [(hidden/32)][4][32]  (b_ih + b_hh)
        const half_t *__restrict__ zeros,   // >= 16 KiB of zeros (h_{-1}, c_{-1})
        float *__restrict__ cbuf,           // [nclusters][KCL][rmax][4][64][4] f32 cell state (private layout)
        unsigned *__restrict__ flags,       // [nclusters][KCL * 4][16]: tiles complete (zeroed per launch)
        unsigned *__restrict__ err,         // [4]: sticky error word
        int T, int N, int reverse, int cpx /* clusters per XCD, 0 = linear map */, int nclusters, int rmax) {
    using L = WsLayout<C>;
    constexpr int KCL = L::KCL, KSX = L::KSX, KS = L::KS, KA = L::KA, NF = L::NF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDSP(unsigned char) smem3 = (LDSP(unsigned char))smem;
    const unsigned lds0 = (unsigned)(size_t)smem3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sim = wave & 3;              // waves w and w + 4 share SIMD w
    const bool hwave = wave >= 4;
    const int l15 = lane & 15, lq = lane >> 4;

    int cl, j;
    {
        const int b = blockIdx.x;
        if (cpx > 0) {
            const int xcd = b & 7, slot = b >> 3;
            cl = xcd * cpx + slot / KCL;
            j = slot % KCL;
        } else {
            cl = b / KCL;
            j = b % KCL;
        }
    }
    if (cl >= nclusters) return;
    const bool xcd_local = cpx > 0;   // every member of the cluster runs on the same XCD (they share one L2)
    const int ntiles = N / WS_TR;
    const int rbase = ntiles / nclusters, rrem = ntiles % nclusters;
    const int R = rbase + (cl < rrem ? 1 : 0);
    const int n0 = (cl * rbase + (cl < rrem ? cl : rrem)) * WS_TR;

    // ---- resident weights: hidden tile jt = 4 j + sim; x-wave k-steps [0, KSX), h-wave [KSX, 2 KSX) ----
    const int jt = j * 4 + sim;
    half8_t w[KSX][4];
    {
        const half_t *wp = Wf16 + ((size_t)jt * KS + (hwave ? KSX : 0)) * 4 * 512 + lane * 8;
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) w[ks][g] = *(const half8_t *)(wp + (ks * 4 + g) * 512);
    }
    if (tid < 256) {
        LDSP(float) bias_s = (LDSP(float))(smem3 + L::OFF_BIAS);
        const int ww = tid >> 6, g = (tid >> 4) & 3, u = tid & 15, jw = j * 4 + ww;
        bias_s[tid] = biasn[((jw >> 1) * 4 + g) * 32 + (jw & 1) * 16 + u];
        LDSP(unsigned) fz = (LDSP(unsigned))(smem3 + L::OFF_FLAGZ);
        fz[tid] = 0u;
    }
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- per-lane constants ----
    // DMA: one instruction fills one k-step block [16 rows][64 B]: lane -> (row = lane / 4, physical 16-byte column
    // lane % 4) <- logical column (lane % 4) ^ ((row >> 2) & 3); the fragment read applies the same XOR.
    const int drow = lane >> 2;
    const unsigned dcol = (unsigned)(((lane & 3) ^ ((drow >> 2) & 3)) * 16);
    const unsigned lane_off = (unsigned)(drow * C * 2) + dcol;   // bytes
    const unsigned lane_off_z = dcol;
    const unsigned foff = (unsigned)(l15 * 64 + ((lq ^ ((l15 >> 2) & 3)) << 4)) + (hwave ? (unsigned)L::OFF_HR : (unsigned)L::OFF_XR);
    const unsigned long long x0 = (unsigned long long)Xin, o0 = (unsigned long long)Xout, z0 = (unsigned long long)zeros;
    const unsigned long long c0 = (unsigned long long)(cbuf + ((((size_t)cl * KCL + j) * rmax) * 4 + sim) * 256);
    gu32 *clflags = (gu32 *)(flags + (size_t)cl * NF * 16);
    gu32 *myflag = (gu32 *)(flags + ((size_t)cl * NF + j * 4 + sim) * 16);
    const unsigned flag_lane = (unsigned)((lane < NF ? lane : 0) * 64);   // bytes
    LDSP(const volatile unsigned) my_fz = (LDSP(const volatile unsigned))(smem3 + L::OFF_FLAGZ + sim * 256) + lane;
    LDSP(const float) my_bias = (LDSP(const float))(smem3 + L::OFF_BIAS) + sim * 64 + 4 * lq;
    // hand-off slot of this SIMD: x-wave -> h-wave [4 gates][64 lanes] float4 accumulators; once the h-wave has taken
    // them it returns, in the same 4 KiB, the new cell state [64 lanes] float4 (gate 0 area) and the 16 x 16 block of
    // h as [16 rows][24 halfs] (gate 1 area) for the x-wave to store (the x-wave owns all global-memory traffic)
    LDSP(unsigned char) hand0 = smem3 + L::OFF_HAND + sim * 4096;              // + slot * 16 KiB
    LDSP(unsigned char) hand = hand0 + lane * 16;                              // + gate * 1 KiB
    bool dead = false;
    const bool tracing = (DBG & 128) && blockIdx.x == 0 && sim == 0 && lane == 0;

    // Tile addresses are running byte offsets (no multiplications in the loop): rel = offset of the tile's 16 rows at
    // its time step, the same in Xin (x_t) and Xout (h_t); h_{t-1} of the tile is Xout + rel - dstep.
    const long long dstep = (reverse ? -1LL : 1LL) * (long long)N * C * 2;   // bytes per time step
    const long long tile_b = (long long)WS_TR * C * 2;                         // bytes per row tile
    const long long wrap_b = dstep - (long long)(R - 1) * tile_b;              // last tile of a step -> first of the next
    const long long rel0 = ((long long)(reverse ? (T - 1) : 0) * N + n0) * C * 2;
    const int total = T * R;

    // DMA requests of the x-waves (wave sim: k-steps sim, sim + 4, ... of a half).  The two halves of a tile live in
    // two rings because they are consumed one iteration apart (the x-waves run one tile ahead): x_t half of tile i + 1 + LA
    // and h_{t-1} half + cell state + flags snapshot of tile i + LA are requested in iteration i.
    auto fetch_x = [&](long long rel, unsigned slot_b) __attribute__((always_inline)) {
        unsigned long long xb = x0 + (unsigned long long)rel;
        asm volatile("" : "+s"(xb));
        const unsigned l = lds0 + L::OFF_XR + slot_b;
#pragma unroll
        for (int q = 0; q < KSX / 4; ++q) {
            const int ks = 4 * q + sim;
            if (DBG & 256) cl_dma16_nt((ghalf_p)(xb + lane_off + (unsigned)(ks * 64)), l + (unsigned)ks * 1024u);
            else if (DBG & 512) cl_dma16((ghalf_p)(xb + lane_off + (unsigned)(ks * 64)), l + (unsigned)ks * 1024u);
            else cl_dma16_sc1((ghalf_p)(xb + lane_off + (unsigned)(ks * 64)), l + (unsigned)ks * 1024u);
        }
    };
    auto fetch_h = [&](long long rel, bool first, unsigned ctile_b, unsigned slot_b) __attribute__((always_inline)) {
        unsigned long long hb = first ? z0 : o0 + (unsigned long long)(rel - dstep);
        unsigned long long cb = first ? z0 : c0 + ctile_b;
        unsigned long long fb = (unsigned long long)clflags;
        asm volatile("" : "+s"(hb));
        asm volatile("" : "+s"(cb));
        asm volatile("" : "+s"(fb));
        const unsigned hoff = first ? lane_off_z : lane_off;
        const unsigned l = lds0 + L::OFF_HR + slot_b;
        cl_dma4_sc1((const unsigned *)(fb + flag_lane), lds0 + L::OFF_FLAGZ + sim * 256);
#pragma unroll
        for (int q = 0; q < KSX / 4; ++q) {
            const int ks = 4 * q + sim;
            cl_dma16_sc1((ghalf_p)(hb + hoff + (unsigned)(ks * 64)), l + (unsigned)ks * 1024u);
        }
        cl_dma16_sc1((ghalf_p)(cb + (unsigned)(lane * 16)), l + KSX * 1024u + (unsigned)sim * 1024u);
    };

    // one half of a tile's MFMAs: KSX k-steps x 4 gates on the fragments of ring slot slot_b
    auto half_tile = [&](float4v_ws (&acc)[4], unsigned slot_b) __attribute__((always_inline)) {
        LDSP(const unsigned char) sp = smem3 + slot_b + foff;
        // fragment ring: 5 buffers, 2 k-steps of look-ahead, order pinned (the keep-alive operands stop the register
        // allocator from folding the ring into fewer physical buffers)
        constexpr int PFD = 5, PLA = 2;
        half8_t bq[PFD];
        if (DBG & 16) {
#pragma unroll
            for (int q = 0; q < PFD; ++q) bq[q] = (half8_t)((half_t)(0.001f * (q + 1)));
        } else {
#pragma unroll
            for (int q = 0; q < PLA; ++q) bq[q] = *(LDSP(const half8_t))(sp + q * 1024);
        }
        __builtin_amdgcn_s_setprio(1);   // the wave in its matrix block wins issue arbitration over its partner's VALU / VMEM block
        cl_static_for<KSX>([&](auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            if (!(DBG & 16) && ks + PLA < KSX) {
                bq[(ks + PLA) % PFD] = *(LDSP(const half8_t))(sp + (ks + PLA) * 1024);
                if (ks > 1) asm volatile("" ::"v"(bq[(ks - 1) % PFD]), "v"(bq[(ks - 2) % PFD]));
                else if (ks > 0) asm volatile("" ::"v"(bq[(ks - 1) % PFD]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(DBG & 4)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (ks < KA)
                        acc[g] = ws_mfma_a(w[ks][g], bq[ks % PFD], acc[g]);
                    else
                        acc[g] = ws_mfma_v(w[ks][g], bq[ks % PFD], acc[g]);
                }
            } else {
                asm volatile("" ::"v"(bq[ks % PFD]));
            }
            __builtin_amdgcn_sched_barrier(0);
        });


        // XDL write -> VALU / LDS-store read of the accumulators: 18 wait states
        __builtin_amdgcn_s_setprio(0);
        // The wait is tied to the accumulators ("+v"): register-only VALU reads of acc could otherwise be scheduled ABOVE a
        // bare asm volatile("s_nop") — right behind the last MFMA, reading accumulators that miss its contribution.
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])::"memory");
    };

    if (!hwave) {
        // =========================== X-WAVES: DMA + x half, one tile ahead ===========================
        // next tiles to request: x half (row tile frx, offset relx, slot) and h half (step fsh, row tile frh, ...)
        int frx = 0, fsh = 0, frh = 0;
        long long relx = rel0, relh = rel0;
        unsigned slotx_b = 0, sloth_b = 0;
        auto advance_x = [&]() __attribute__((always_inline)) {
            if (++frx == R) {
                frx = 0;
                relx += wrap_b;
            } else {
                relx += tile_b;
            }
        };
        auto advance_h = [&]() __attribute__((always_inline)) {
            if (++frh == R) {
                frh = 0;
                ++fsh;
                relh += wrap_b;
            } else {
                relh += tile_b;
            }
        };
#pragma unroll
        for (int d = 0; d <= WS_LA; ++d) {   // x halves of tiles 0 .. LA, h halves of tiles 0 .. LA - 1
            fetch_x(relx, slotx_b);
            slotx_b = (slotx_b + L::XS == WS_NSL * L::XS) ? 0u : slotx_b + L::XS;
            advance_x();
            if (d < WS_LA) {
                fetch_h(relh, true, 0u, sloth_b);
                sloth_b = (sloth_b + L::HS == WS_NSL * L::HS) ? 0u : sloth_b + L::HS;
                advance_h();
            }
        }
        unsigned cslot_b = 0;               // ring slot of the tile whose x half is computed next
        int pr = 0;                         // row tile / byte offset of the tile whose results are stored next
        long long prel = rel0;
        // x half of tile 0 (hand-off slot 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            float4v_ws acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = *(LDSP(const float4v_ws))(my_bias + g * 16);
            half_tile(acc, cslot_b);
#pragma unroll
            for (int g = 0; g < 4; ++g) *(LDSP(float4v_ws))(hand + g * 1024) = acc[g];
            cslot_b += L::XS;
        }
#pragma nounroll
        for (int i = 0; i <= total; ++i) {   // iteration `total` only stores the last tile's results
            WS_STAMP(0, 5);
            // tile i + 1 (x half, this wave) and tile i (h half, partner) have landed: both requested in iteration
            // i - LA; younger requests = the XOPS of each of the LA - 1 iterations since
            if ((DBG & 2) || i <= WS_LA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::XOPS * (WS_LA - 1)) : "memory");
            }
            WS_STAMP(0, 6);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own hand-off stores of the previous iteration
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WS_STAMP(0, 0);
            // ---- publish: the wait above retired every operation of iterations <= i - LA, i.e. the stores of tiles
            // 0 .. i - LA - 1 ----
            if (!(DBG & 32) && lane == 0) {
                const unsigned p = (unsigned)(i > WS_LA ? i - WS_LA : 0);
                if (xcd_local)   // members share one L2: a plain (write-back) store is visible to their sc1 loads
                    __hip_atomic_store(myflag, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_store(myflag, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ---- results of tile i - 1 (left in the hand-off slot by the partner): h block and cell state ----
            {
                LDSP(const unsigned char) hb = hand0 + (((i - 1) & 1) ? 16384 : 0);
                const int prow = lane >> 1, seg = lane & 1;
                const half8_t v = *(LDSP(const half8_t))(hb + 1024 + (prow * 24 + seg * 8) * 2);
                const float4v_ws cn = *(LDSP(const float4v_ws))(hb + lane * 16);
                const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(o0 + (unsigned long long)prel), 0, WS_TR * C * 2, 0x00020000);
                if (i > 0 && !(DBG & 32)) {
                    const auto vv = __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v);
                    const int so = (lane < 32) ? (prow * C + j * 64 + sim * 16 + seg * 8) * 2 : WS_TR * C * 2;   // >= 32: out of range, dropped
                    if (xcd_local)
                        __builtin_amdgcn_raw_buffer_store_b128(vv, ors, so, 0, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(vv, ors, so, 0, 16 /* sc1: write-through */);
                    // cell state back to its private tile (read again one time step = R tiles later)
                    *(float4v_ws *)(c0 + (size_t)pr * 4096 + (unsigned)(lane * 16)) = cn;
                    if (++pr == R) {
                        pr = 0;
                        prel += wrap_b;
                    } else {
                        prel += tile_b;
                    }
                }
            }
            WS_STAMP(0, 1);
            // ---- requests: h half of tile f = i + LA (its h rows were produced as tile f - R by all members), x half of
            // tile f + 1 ----
            {
                const int f = i + WS_LA;
                if (f < total && fsh > 0 && !(DBG & 8) && !dead) {
                    const unsigned need = (unsigned)(f - R + 1);
                    const unsigned snap = *my_fz;
                    if (!__all(lane >= NF || snap >= need)) {
#ifdef MIBC_DEBUG_KERNELS
                        if (lane == 0) atomicAdd(err + 1, 1u);           // slow-path entries
                        if (lane < NF && snap < need) atomicMax(err + 2, need - snap);   // worst lag seen
#endif
                        // slow path: refresh the snapshot by DMA and re-read it (a drain is harmless for the counted
                        // waits: they only rely on issue order)
                        unsigned spins = 0;
                        bool good;
                        do {
                            cl_dma4_sc1((const unsigned *)((unsigned long long)clflags + flag_lane),
                                        lds0 + L::OFF_FLAGZ + sim * 256);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            const unsigned v = *my_fz;
                            good = __all(lane >= NF || v >= need);
                            if (!good) __builtin_amdgcn_s_sleep(8);
                        } while (!good && ++spins < WS_SPIN_LIMIT);
                        if (!good) {
                            dead = true;
                            if (lane == 0) atomicCAS(err, 0u, 0x80000000u | ((unsigned)cl << 16) | (unsigned)(fsh & 0xffff));
                        }
                    }
                }
                if (DBG & 64) {   // ablation: every request hits the same cache-resident lines
                    fetch_h(rel0, true, 0u, sloth_b);
                    fetch_x(rel0, slotx_b);
                } else if (!(DBG & 2) || i < R) {
                    fetch_h(relh, fsh == 0, (unsigned)frh * 4096u, sloth_b);
                    fetch_x(relx, slotx_b);
                }
                sloth_b = (sloth_b + L::HS == WS_NSL * L::HS) ? 0u : sloth_b + L::HS;
                slotx_b = (slotx_b + L::XS == WS_NSL * L::XS) ? 0u : slotx_b + L::XS;
                // past the end the last tile is requested again (nobody reads it)
                if (f + 1 < total) advance_h();
                if (f + 2 < total) advance_x();
            }
            WS_STAMP(0, 2);
            // ---- x half of tile i + 1 -> hand-off slot (i + 1) & 1 (past the end: a dummy pass, nobody reads it) ----
            float4v_ws acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = *(LDSP(const float4v_ws))(my_bias + g * 16);
            half_tile(acc, cslot_b);
            WS_STAMP(0, 3);
            LDSP(unsigned char) hd = hand + (((i + 1) & 1) ? 16384 : 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) *(LDSP(float4v_ws))(hd + g * 1024) = acc[g];
            cslot_b = (cslot_b + L::XS == WS_NSL * L::XS) ? 0u : cslot_b + L::XS;
            WS_STAMP(0, 4);
        }
    } else {
        // =========================== H-WAVES: h half, gates (no global-memory traffic) ===========================
        unsigned pslot_b = 0;
        __builtin_amdgcn_s_barrier();       // pairs with the x-waves' prologue barrier
        asm volatile("" ::: "memory");
#pragma nounroll
        for (int i = 0; i <= total; ++i) {
            WS_STAMP(1, 5);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own result stores of the previous iteration
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i == total) break;
            WS_STAMP(1, 0);
            // ---- continue the partner's accumulators with the h half of tile i ----
            float4v_ws acc[4];
            LDSP(unsigned char) hs = hand0 + ((i & 1) ? 16384 : 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = *(LDSP(const float4v_ws))(hs + lane * 16 + g * 1024);
            half_tile(acc, pslot_b);
            const float4v_ws cv = *(LDSP(const float4v_ws))(smem3 + L::OFF_HR + pslot_b + KSX * 1024 + sim * 1024 + lane * 16);
            WS_STAMP(1, 2);
            // ---- gates (D row = hidden 4 lq + e, D col = batch row l15) ----
            float4v_ws cn;
            half4_t hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (DBG & 1) {
                    cn[e] = cv[e];
                    hv[e] = (half_t)(1e-3f * (acc[0][e] + acc[1][e] + acc[2][e] + acc[3][e]));
                } else {
                    const float ig = fast_sigmoid(acc[0][e]);
                    const float fg = fast_sigmoid(acc[1][e]);
                    const float gg = fast_tanh(acc[2][e]);
                    const float og = fast_sigmoid(acc[3][e]);
                    const float c = fmaf(fg, cv[e], ig * gg);
                    cn[e] = c;
                    hv[e] = (half_t)(og * fast_tanh(c));
                }
            }
            // results into the (consumed) hand-off slot; the partner stores them in the next iteration
            *(LDSP(float4v_ws))(hs + lane * 16) = cn;
            *(LDSP(half4_t))(hs + 1024 + (l15 * 24 + 4 * lq) * 2) = hv;
            WS_STAMP(1, 4);
            pslot_b = (pslot_b + L::HS == WS_NSL * L::HS) ? 0u : pslot_b + L::HS;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

extern "C" size_t mibc_lstm_ws_lds_bytes(int C) { return C == 384 ? (size_t)WsLayout<384>::BYTES : 0; }

// geometry of a launch: clusters, row tiles per cluster (max), clusters per XCD (0 = linear map); 0 if not covered
static int ws_geometry(int C, int N, int *nclusters, int *rmax, int *cpx) {
    if (C != 384 || N % WS_TR != 0) return 0;
    static int ncu = 0;
    if (ncu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int KCL = C / 64;
    const int ntiles = N / WS_TR;
    const int per_xcd = (ncu / 8) / KCL;          // clusters whose members share one XCD
    int ncl = per_xcd * 8;
    if (ncl < 1) return 0;
    if (ntiles / ncl < WS_RMIN) ncl = ntiles / WS_RMIN;
    if (ncl < 1) return 0;
    *nclusters = ncl;
    *rmax = (ntiles + ncl - 1) / ncl;
    *cpx = (ncl % 8 == 0) ? ncl / 8 : 0;
    return 1;
}

extern "C" size_t mibc_lstm_ws_cstate_bytes(int C, int N) {
    int ncl, rmax, cpx;
    if (!ws_geometry(C, N, &ncl, &rmax, &cpx)) return 0;
    return (size_t)ncl * (C / 64) * rmax * 4096;
}
extern "C" size_t mibc_lstm_ws_flag_bytes(int C, int N) {
    int ncl, rmax, cpx;
    if (!ws_geometry(C, N, &ncl, &rmax, &cpx)) return 0;
    return (size_t)ncl * (C / 64) * 4 * 16 * sizeof(unsigned);
}

// Returns 0 if launched, 1 if the shape is not covered (caller falls back to lstm_layer_x8_kernel).
extern "C" int mibc_launch_lstm_layer_ws(hipStream_t s, int C, const half_t *Xin, half_t *Xout, const half_t *Wf16,
                                         const float *biasn, const half_t *zeros, float *cbuf, unsigned *flags,
                                         unsigned *err, int T, int N, int reverse) {
    if (!Wf16 || !biasn || !zeros || !cbuf || !flags || !err) return 1;
    int ncl, rmax, cpx;
    if (!ws_geometry(C, N, &ncl, &rmax, &cpx)) return 1;
    using L = WsLayout<384>;
    const dim3 grid(ncl * L::KCL);
    if (hipMemsetAsync(flags, 0, mibc_lstm_ws_flag_bytes(C, N), s) != hipSuccess) return 1;
#define WS_LAUNCH(D_)                                                                                              \
    do {                                                                                                           \
        static bool once = false;                                                                                  \
        if (!once) {                                                                                               \
            (void)hipFuncSetAttribute((const void *)lstm_layer_ws_kernel<384, D_>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);                       \
            once = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL((lstm_layer_ws_kernel<384, D_>), grid, dim3(512), L::BYTES, s, Xin, Xout, Wf16, biasn, \
                           zeros, cbuf, flags, err, T, N, reverse, cpx, ncl, rmax);                                \
    } while (0)
#ifdef MIBC_DEBUG_KERNELS
    static const int dbg = MIBC_ENV_INT("MIBC_WS_LSTM_DBG", 0);
    switch (dbg) {
        case 1: WS_LAUNCH(1); return 0;
        case 2: WS_LAUNCH(2); return 0;
        case 4: WS_LAUNCH(4); return 0;
        case 8: WS_LAUNCH(8); return 0;
        case 16: WS_LAUNCH(16); return 0;
        case 18: WS_LAUNCH(18); return 0;
        case 19: WS_LAUNCH(19); return 0;
        case 27: WS_LAUNCH(27); return 0;
        case 6: WS_LAUNCH(6); return 0;
        case 14: WS_LAUNCH(14); return 0;
        case 46: WS_LAUNCH(46); return 0;
        case 59: WS_LAUNCH(59); return 0;
        case 63: WS_LAUNCH(63); return 0;
        case 10: WS_LAUNCH(10); return 0;
        case 42: WS_LAUNCH(42); return 0;
        case 256: WS_LAUNCH(256); return 0;
        case 512: WS_LAUNCH(512); return 0;
        case 128: WS_LAUNCH(128); return 0;
        case 192: WS_LAUNCH(192); return 0;
        case 64: WS_LAUNCH(64); return 0;
        case 72: WS_LAUNCH(72); return 0;
        case 68: WS_LAUNCH(68); return 0;
        case 32: WS_LAUNCH(32); return 0;
        case 40: WS_LAUNCH(40); return 0;
        case 36: WS_LAUNCH(36); return 0;
        default: break;
    }
#endif
    WS_LAUNCH(0);
#undef WS_LAUNCH
    return 0;
}
