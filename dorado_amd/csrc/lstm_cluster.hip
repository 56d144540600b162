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
// 8 waves = 4 row groups x 2 hidden groups on v_mfma_f32_32x32x16_f16.  The cell state stays in
// registers (fp32, 64 per lane); gates are lane-local exactly as in lstm.hip.
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
#include "common.h"

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
#define CL_OFF_FLAGZ (CL_OFF_BIAS + 2 * 2 * 4 * 32 * 4)
#define CL_OFF_SYNC (CL_OFF_FLAGZ + 256)
#define CL_LDS_BYTES (CL_OFF_SYNC + 64)

typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void cl_dma16(const half_t *g, half_t *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
__device__ __forceinline__ void cl_dma16_sc1(const half_t *g, half_t *l) {   // agent-scope (L1 bypass) load
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 16);
}
__device__ __forceinline__ void cl_dma4_sc1(const unsigned *g, unsigned *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 4, 0, 16);
}

// One launch = one layer.  grid = KCL * (clusters resident at once); a workgroup loops over the row
// groups (clusters of 256 rows) rg = first, first + stride, ... so that N may exceed one residency.
template <int C, bool MASKED>
__global__ __launch_bounds__(512, 2) void lstm_layer_cl_kernel(
        const half_t *__restrict__ Xin,     // [T][N][C]
        half_t *__restrict__ Xout,          // [T][N][C]
        const half_t *__restrict__ Wt,      // [KCL][2][KS][256][32]: swizzled LDS images of the weight slabs
        const float *__restrict__ biascl,   // [KCL][2][2][4][32]  (b_ih + b_hh)
        const half_t *__restrict__ zeros,   // [256][C] zeros (h_{-1})
        unsigned *__restrict__ flags,       // [nclusters][KCL][16]: completed steps of member j (zeroed per launch)
        unsigned *__restrict__ err,         // [4]: sticky error word, first failing (cluster, step)
        int T, int N, int reverse, int cpx /* clusters per XCD slot group, 0 = linear map */,
        int resident_clusters,
        const unsigned long long *__restrict__ tmask /* MASKED: [T][N/64] */) {
    constexpr int KCL = C / 128;
    constexpr int KSX = C / CL_BK;         // x-part slabs per pass
    constexpr int KS = 2 * KSX;            // slabs per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t *stage = (half_t *)smem;
    half_t *patch_all = (half_t *)(smem + CL_OFF_PATCH);
    float *bias_s = (float *)(smem + CL_OFF_BIAS);
    unsigned *flagz = (unsigned *)(smem + CL_OFF_FLAGZ);
    volatile unsigned *syncw = (volatile unsigned *)(smem + CL_OFF_SYNC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
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
    // row = (wave*2+q)*16 + lane/4, physical 16-byte column lane%4 <- logical column (lane%4) ^ ((row>>2)&3)
    unsigned aoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        const int col = (lane & 3) ^ ((row >> 2) & 3);
        aoff[q] = (unsigned)(row * C + col * 8);
    }
    const half_t *wsrc = Wt + (size_t)j * 2 * KS * CL_WTILE + (size_t)(wave * 2) * 512 + lane * 8;

    for (int i = tid; i < 2 * 2 * 4 * 32; i += 512) bias_s[i] = biascl[(size_t)j * 2 * 2 * 4 * 32 + i];
    half_t *patch = patch_all + wave * CL_PATCH;
    bool dead = false;

    for (int cl = cl0; cl < nclusters; cl += resident_clusters) {
        const int n0 = cl * CL_ROWS;
        unsigned *myflag = flags + ((size_t)cl * KCL + j) * 16;
        const unsigned *clflags = flags + (size_t)cl * KCL * 16;

        float16_t cst[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) cst[a][b][r] = 0.0f;

        // ---- slab stream state of the ISSUE side (runs 3 slabs ahead of the compute side) ----
        int is_step = 0, is_u = 0;      // next slab to issue: step, u = pass * KS + ks
        int ring_w = 0;                 // ring slot it goes to
        auto issue_next = [&]() {
            if (is_step >= T) return;
            const int t = reverse ? (T - 1 - is_step) : is_step;
            const int p = is_u / KS, ks = is_u % KS;
            half_t *Ws = stage + ring_w * CL_STAGE, *As = Ws + CL_WTILE;
            const half_t *wb = wsrc + (size_t)(p * KS + ks) * CL_WTILE;
            cl_dma16(wb, Ws + (wave * 2) * 512);
            cl_dma16(wb + 512, Ws + (wave * 2 + 1) * 512);
            if (ks < KSX) {
                const half_t *xb = Xin + ((size_t)t * N + n0) * C + ks * CL_BK;
                cl_dma16(xb + aoff[0], As + (wave * 2) * 512);
                cl_dma16(xb + aoff[1], As + (wave * 2 + 1) * 512);
            } else {
                const int tp = reverse ? (t + 1) : (t - 1);
                const half_t *hb = (is_step == 0) ? zeros : (Xout + ((size_t)tp * N + n0) * C);
                hb += (ks - KSX) * CL_BK;
                cl_dma16_sc1(hb + aoff[0], As + (wave * 2) * 512);
                cl_dma16_sc1(hb + aoff[1], As + (wave * 2 + 1) * 512);
            }
            ring_w = (ring_w + 1) & (CL_NST - 1);
            if (++is_u == 2 * KS) {
                is_u = 0;
                ++is_step;
            }
        };
        __syncthreads();   // bias_s visible; previous row group's LDS reads are done
        issue_next();
        issue_next();
        issue_next();
        int ring_r = 0;

        for (int step = 0; step < T; ++step) {
            const int t = reverse ? (T - 1 - step) : step;
            const bool last_step = (step == T - 1);
            unsigned long long vm = ~0ull;
            if (MASKED) vm = tmask[(size_t)t * (N / 64) + (n0 >> 6) + rgw];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float16_t acc[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float *bp = bias_s + ((p * 2 + hg) * 4 + g) * 32 + 4 * lhi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4_t v = *(const float4_t *)(bp + 8 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[g][0][q * 4 + e] = v[e];
                            acc[g][1][q * 4 + e] = v[e];
                        }
                    }
                }
#pragma nounroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int u = p * KS + ks;
                    // slab (step, u) has landed when only the two younger slabs (8 DMAs per lane) are still in
                    // flight; extra younger operations (h stores, flag fetch) only make this wait stricter
                    const bool tail = last_step && (u >= 2 * KS - 2);
                    if (p == 0 && ks == 2 && step > 0) {
                        // publish h_{step-1}: the sc1 stores of both passes were issued >= 2 slabs ago
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else if (!tail) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    if (p == 0 && step > 0) {
                        if (ks == 2) {
                            if (tid == 0 && !dead)
                                __hip_atomic_store((gu32 *)myflag, (unsigned)step, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                        } else if (ks == KSX - 6) {
                            if (wave == 0) cl_dma4_sc1(clflags + (lane < KCL ? lane : 0) * 16, flagz);
                        } else if (ks == KSX - 3 && !dead) {
                            // h_{step-1} of every member must be visible before the first h slab is fetched
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
                                    if (tid == 0) {
                                        atomicCAS(err, 0u, 0x80000000u | ((unsigned)cl << 16) | (unsigned)step);
                                    }
                                }
                                __syncthreads();
                            }
                        }
                    }
                    issue_next();
                    const half_t *Ws = stage + ring_r * CL_STAGE, *As = Ws + CL_WTILE;
                    ring_r = (ring_r + 1) & (CL_NST - 1);
#pragma unroll
                    for (int k16 = 0; k16 < 2; ++k16) {
                        half8_t wf[4], xf[2];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int row = hg * 128 + g * 32 + l31;
                            wf[g] = *(const half8_t *)(Ws + row * CL_BK + (((2 * k16 + lhi) ^ ((row >> 2) & 3)) << 3));
                        }
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            const int row = rgw * 64 + rt * 32 + l31;
                            xf[rt] = *(const half8_t *)(As + row * CL_BK + (((2 * k16 + lhi) ^ ((row >> 2) & 3)) << 3));
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt) acc[g][rt] = mfma32x32x16(wf[g], xf[rt], acc[g][rt]);
                    }
                }
                // ---- gates of this pass: D row = hidden (r&3) + 8 (r>>2) + 4 lhi, D col = batch row l31 ----
                const int hcol = j * 128 + p * 64 + hg * 32;
                const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(Xout + ((size_t)t * N + n0 + rgw * 64) * C + hcol), 0, 64 * C * 2, 0x00020000);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const bool rowon = !MASKED || ((vm >> (rt * 32 + l31)) & 1ull);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        half4_t hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = q * 4 + e;
                            const float ig = fast_sigmoid(acc[0][rt][r]);
                            const float fg = fast_sigmoid(acc[1][rt][r]);
                            const float gg = fast_tanh(acc[2][rt][r]);
                            const float og = fast_sigmoid(acc[3][rt][r]);
                            float c = fmaf(fg, cst[p][rt][r], ig * gg);
                            float hval = og * fast_tanh(c);
                            if (MASKED && !rowon) {
                                c = 0.0f;
                                hval = 0.0f;
                            }
                            cst[p][rt][r] = c;
                            hv[e] = (half_t)hval;
                        }
                        *(half4_t *)(patch + l31 * CL_PATCH_LD + 8 * q + 4 * lhi) = hv;
                    }
                    __builtin_amdgcn_wave_barrier();   // same wave: LDS operations execute in order
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int prow = (lane >> 2) + 16 * i, seg = lane & 3;
                        const half8_t v = *(const half8_t *)(patch + prow * CL_PATCH_LD + seg * 8);
                        __builtin_amdgcn_raw_buffer_store_b128(
                                __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), ors,
                                ((rt * 32 + prow) * C + seg * 8) * 2, 0, 16 /* sc1: write-through */);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

extern "C" size_t mibc_lstm_cl_lds_bytes(void) { return CL_LDS_BYTES; }

// Returns 0 if launched, 1 if the shape is not covered (caller falls back to the per-workgroup kernels).
extern "C" int mibc_launch_lstm_layer_cl(hipStream_t s, int C, const half_t *Xin, half_t *Xout, const half_t *Wt,
                                         const float *biascl, const half_t *zeros, unsigned *flags, unsigned *err,
                                         int T, int N, int reverse, const unsigned long long *tmask) {
    if (Wt == nullptr || biascl == nullptr || zeros == nullptr || flags == nullptr || err == nullptr) return 1;
    if ((C != 512 && C != 768 && C != 1024) || N < CL_ROWS || N % CL_ROWS != 0) return 1;
    const int KCL = C / 128;
    const int nclusters = N / CL_ROWS;
    static int ncu = 0;
    if (ncu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    // every member of a cluster must be resident at the same time: one workgroup per CU (the LDS request
    // guarantees it), never more workgroups than CUs
    int resident = ncu / KCL;
    if (resident < 1) return 1;
    if (resident > nclusters) resident = nclusters;
    int cpx = 0;
    if (resident % 8 == 0 && (resident / 8) * KCL * 8 <= ncu) cpx = resident / 8;   // same-XCD clusters
    const dim3 grid(resident * KCL);
    if (hipMemsetAsync(flags, 0, (size_t)nclusters * KCL * 16 * sizeof(unsigned), s) != hipSuccess) return 1;
#define CL_LAUNCH(CC, M_)                                                                                   \
    do {                                                                                                    \
        static bool once = false;                                                                           \
        if (!once) {                                                                                        \
            (void)hipFuncSetAttribute((const void *)lstm_layer_cl_kernel<CC, M_>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, CL_LDS_BYTES);            \
            once = true;                                                                                    \
        }                                                                                                   \
        hipLaunchKernelGGL((lstm_layer_cl_kernel<CC, M_>), grid, dim3(512), CL_LDS_BYTES, s, Xin, Xout, Wt, \
                           biascl, zeros, flags, err, T, N, reverse, cpx, resident, tmask);                 \
    } while (0)
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
}
