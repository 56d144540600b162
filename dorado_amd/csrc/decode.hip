// dorado_amd/csrc/decode.hip — CRF decoder on the GPU.
//
// Replaces the reference's CPU decoder (dorado/basecall/decode/CPUDecoder.cpp:17-157 +
// decode/beam_search.cpp:54-520) and the four closed Koi launches of the CUDA path
// (decode/CUDADecoder.cpp:76-100: back_guide, beam_search, compute_posts, run_decode) with three
// kernels.  Compile with -ffp-contract=off: the scan / beam arithmetic is a fixed IEEE operation
// sequence (detmath.h) so that moves and bases are bit-comparable with the CPU oracle.
//
//   k1 bwd_scan_kernel   one workgroup per chunk, one thread per state, T sequential steps:
//                        beta[t][s] = LSE(beta[t+1][s]+stay, beta[t+1][succ_b(s)]+M[t][succ_b(s),s])
//                        state vector in LDS (ping-pong), score row prefetched one step ahead and
//                        staged transposed in LDS ([base][state]) so every read is conflict-free.
//                        HBM: reads 2K B/step (scores), writes 4S B/step (back-guides).
//   k2 beam_search64_kernel one WAVE per chunk: beam (width <= 32) in registers (both wave halves
//                        hold it and share the expansion and the merge test), 5W candidates in
//                        LDS, hash-merge / bisection cut-off / in-order compaction with wave
//                        ballots; trace (4 B x W per step) to HBM, traced back through LDS tiles.
//                        (beam_search_kernel = the 32-lane form of rounds 1-5: debug library only.)
//   k3 posts_qual_kernel one workgroup per chunk: forward scan fused with the posterior of the
//                        called k-mer (+ its shifted neighbours) — posts[T+1][S] never touches
//                        HBM —, then sequence / qstring emission.
#include "common.h"
#include "detmath.h"
#include <stdlib.h>

#define FLT_LOWEST (-3.402823466e+38f)

// Variable-chunk mode: "chunk" n of a launch is an arbitrary step interval of the packed
// [N][T] planes instead of a whole row: so = first step (offset into scores / path / output planes),
// bo = first row of its back-guide / trace block (T_n + 1 rows), T_n = its length.  All null = one chunk per row.
struct VarIdx {
    const int *coff, *boff, *clen;
};
#define VAR_SETUP(T_)                                                       \
    size_t so = (size_t)n * (T_), bo = (size_t)n * ((T_) + 1);              \
    if (vi.coff != nullptr) {                                               \
        so = (size_t)vi.coff[n];                                            \
        bo = (size_t)vi.boff[n];                                            \
        T_ = vi.clen[n];                                                    \
    }

__device__ __forceinline__ float clampf(float v, float c) {
    return (c > 0.0f) ? fminf(fmaxf(v, -c), c) : v;
}

// ---------------------------------------------------------------------------------------------
// k1: backward scan (decode/CPUDecoder.cpp:66-92 + scan :17-38)
// ---------------------------------------------------------------------------------------------
__global__ void bwd_scan_kernel(const half_t *__restrict__ scores,  // [N][T][4S]
                                float *__restrict__ bwd,            // [N][T+1][S]
                                int T, int S, float stay, float clampv, VarIdx vi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *beta = (float *)smem;                     // [2][S]
    half_t *sc = (half_t *)(smem + 2 * S * 4);       // [2][4][S]  ([buf][base that fell off][dest])
    const int n = blockIdx.x;
    const int s = threadIdx.x;
    const int K = 4 * S;
    const int Q = S >> 2;
    VAR_SETUP(T)
    const half_t *sn = scores + so * K;
    float *bn = bwd + bo * S;

    beta[s] = 0.0f;
    bn[(size_t)T * S + s] = 0.0f;
    float mine = 0.0f;
    half4_t row = *(const half4_t *)(sn + (size_t)(T - 1) * K + 4 * s);
    const int hi = s / Q;
    const int n0 = (s << 2) & (S - 1);
    const half_t ch = (half_t)((clampv > 0.0f) ? fminf(clampv, 65504.0f) : 65504.0f);
    int p = 0;
    for (int t = T - 1; t >= 0; --t) {
        half_t *scw = sc + p * K;
        // element (dest = s, hi = b) of row t  ->  scw[b][s]
        scw[0 * S + s] = row[0];
        scw[1 * S + s] = row[1];
        scw[2 * S + s] = row[2];
        scw[3 * S + s] = row[3];
        __syncthreads();
        if (t > 0) {
            row = *(const half4_t *)(sn + (size_t)(t - 1) * K + 4 * s);
        }
        const float4_t b4 = *(const float4_t *)(beta + p * S + n0);
        // clamp in f16 (packed; +-5 is exact in f16, so this equals clamping the converted floats)
        const half4_t m4 = __builtin_elementwise_min(
                __builtin_elementwise_max(*(const half4_t *)(scw + hi * S + n0), (half4_t)(-ch)), (half4_t)(ch));
        const float v = dm_lse5(mine + stay, b4[0] + (float)m4[0], b4[1] + (float)m4[1],
                                b4[2] + (float)m4[2], b4[3] + (float)m4[3]);
        mine = v;
        beta[(p ^ 1) * S + s] = v;
        bn[(size_t)t * S + s] = v;
        p ^= 1;
    }
}

// k1 (two states per thread): thread j owns states j and j + S/2.  Both read the SAME four successor back-guides
// (their successor index (s << 2) mod S coincides) and differ only in the score row (fell-off base hi and hi + 2),
// so the log-sum-exp runs on the packed-f32 pipe for two states at once and the LDS guide reads are halved.
// Element-wise the arithmetic is the scalar kernel's: bit-identical output.
__global__ void bwd_scan2_kernel(const half_t *__restrict__ scores,  // [N][T][4S]
                                 float *__restrict__ bwd,            // [N][T+1][S]
                                 int T, int S, float stay, float clampv, VarIdx vi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *beta = (float *)smem;                     // [2][S]
    half_t *sc = (half_t *)(smem + 2 * S * 4);       // [2][4][S]  ([buf][base that fell off][dest])
    const int n = blockIdx.x;
    const int j = threadIdx.x;                       // 0 .. S/2 - 1
    const int H = S >> 1;
    const int K = 4 * S;
    const int Q = S >> 2;
    VAR_SETUP(T)
    const half_t *sn = scores + so * K;
    float *bn = bwd + bo * S;

    beta[j] = 0.0f;
    beta[j + H] = 0.0f;
    bn[(size_t)T * S + j] = 0.0f;
    bn[(size_t)T * S + j + H] = 0.0f;
    dm_f2 mine = (dm_f2)(0.0f);
    half8_t row = *(const half8_t *)(sn + (size_t)(T - 1) * K + 8 * j);   // scores of dest states 2j, 2j+1
    const int hi = j / Q;                            // fell-off base of state j; state j + S/2 has hi + 2
    const int n0 = (j << 2) & (S - 1);
    const half_t ch = (half_t)((clampv > 0.0f) ? fminf(clampv, 65504.0f) : 65504.0f);
    typedef _Float16 half2_l __attribute__((ext_vector_type(2)));
    int p = 0;
    for (int t = T - 1; t >= 0; --t) {
        half_t *scw = sc + p * K;
#pragma unroll
        for (int b = 0; b < 4; ++b) *(half2_l *)(scw + b * S + 2 * j) = half2_l{row[b], row[4 + b]};
        __syncthreads();
        if (t > 0) {
            row = *(const half8_t *)(sn + (size_t)(t - 1) * K + 8 * j);
        }
        const float4_t b4 = *(const float4_t *)(beta + p * S + n0);
        const half4_t ma = __builtin_elementwise_min(
                __builtin_elementwise_max(*(const half4_t *)(scw + hi * S + n0), (half4_t)(-ch)), (half4_t)(ch));
        const half4_t mb = __builtin_elementwise_min(
                __builtin_elementwise_max(*(const half4_t *)(scw + (hi + 2) * S + n0), (half4_t)(-ch)), (half4_t)(ch));
        const dm_f2 v = dm2_lse5(mine + (dm_f2)(stay), dm_f2{b4[0] + (float)ma[0], b4[0] + (float)mb[0]},
                                 dm_f2{b4[1] + (float)ma[1], b4[1] + (float)mb[1]},
                                 dm_f2{b4[2] + (float)ma[2], b4[2] + (float)mb[2]},
                                 dm_f2{b4[3] + (float)ma[3], b4[3] + (float)mb[3]});
        mine = v;
        beta[(p ^ 1) * S + j] = v[0];
        beta[(p ^ 1) * S + j + H] = v[1];
        bn[(size_t)t * S + j] = v[0];
        bn[(size_t)t * S + j + H] = v[1];
        p ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// k2: beam search (decode/beam_search.cpp:125-455), one wave per chunk
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t crc32c_bits(uint32_t crc, uint32_t bits, int nbits) {
    // beam_search.cpp:104-121 (reversed Castagnoli polynomial, LSB first)
    for (int i = 0; i < nbits; ++i) {
        const uint32_t b = (bits ^ crc) & 1u;
        crc >>= 1;
        if (b) crc ^= 0x82f63b78u;
        bits >>= 1;
    }
    return crc;
}
// Wave-wide reductions on the DPP network (no LDS round trips): four steps inside every row of 16
// lanes, then row_bcast:15 / row_bcast:31 fold the four rows into lane 63, which is broadcast
// through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float oldv, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv),
                                                                 __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
// (v_max_f32_dpp / v_add_f32_dpp: one instruction per stage instead of v_mov_dpp + canonicalise + op; same operands, max and
//  add commute, so the results are those of the two-instruction forms bit for bit.  wave_sum: rows outside a stage's row_mask
//  keep their value, which equals adding the +0 the two-instruction form moved in — its users sum non-negative terms.
//  s_nop 1: a VGPR written by a VALU instruction is read through DPP no sooner than two wait states later; the last one
//  covers the v_readlane.)
__device__ __forceinline__ float wave_max(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"      // every lane holds its row's maximum
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"    // into rows 1 and 3
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"    // into rows 2 and 3
        "s_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {  // summation order is fixed but not the butterfly's
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t f2key(float f) {  // monotone float -> uint
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int lanes_below(unsigned long long m, int lane) {
    return __popcll(m & ((1ull << lane) - 1ull));
}

#define BS_MAXW 32
#define BS_CAND (5 * BS_MAXW)

// HT (round 5): the candidates' sequence hashes are kept BASE-major ([newest base b][beam element e], stays behind them), so that
// the stay / step merge test — "is there a step with my newest base and my hash" — reads its 32 candidates as eight
// ds_read_b128 of one row instead of 32 dependent ds_read_b32 in a loop of run-time length (the largest part of the ~450
// instruction step).  Same matches, same order, same outputs; HT = false is the round 1-4 loop (debug build, MIBC_K2_HT=0).
template <int S, bool HT = true>
// amdgpu_waves_per_eu(8): the kernel is latency-bound (one wave per chunk, a serial chain per step), so resident waves are its
// throughput; left alone hipcc takes 86 VGPRs for S = 256 (5 waves per SIMD = 20 chunks per CU); asked for 8 it needs 63 without
// scratch, and the 5.5 KB LDS arena then sets the limit (29 chunks per CU).  (S = 1024 keeps its 138 registers: LDS-bound at 10.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void beam_search_kernel(
        const half_t *__restrict__ scores,   // [N][T][4S]
        const float *__restrict__ bwd,       // [N][T+1][S]
        uint32_t *__restrict__ trace,        // [N][T+1][W]  state | prev<<16 | stay<<24
        uint16_t *__restrict__ path_state,   // [N][T]  full k-mer state per block
        int8_t *__restrict__ moves,          // [N][T]
        int T, int W, float log_cut, float stay, float clampv, VarIdx vi) {
    constexpr int K = 4 * S;
    constexpr int BITS = (S == 64) ? 6 : (S == 256) ? 8 : (S == 1024) ? 10 : 12;
    constexpr int RPL = (K / 64 / 8) > 0 ? (K / 64 / 8) : 1;  // half8 loads per lane per score row
    constexpr int GPL = S / 64;      // floats per lane for one guide row
    // one LDS arena: the search phase (score row, guide row, 5W candidates, new front) and the
    // trace-back phase (32-row tile of the beam trace) never overlap in time
    constexpr int A_SC = 0;                              // half_t sc_row[K]
    constexpr int A_BG = A_SC + K * 2;                   // float bg_row[S]
    constexpr int A_CS = A_BG + S * 4;                   // float c_score[BS_CAND]
    constexpr int A_CH = A_CS + BS_CAND * 4;             // uint32 c_hash[BS_CAND]
    constexpr int A_CT = A_CH + BS_CAND * 4;             // uint16 c_state[BS_CAND]
    constexpr int A_TG = A_CT + BS_CAND * 2;             // int tag[4W]
    constexpr int A_NS = A_TG + 4 * BS_MAXW * 4;         // float n_score[W]
    constexpr int A_NH = A_NS + BS_MAXW * 4;             // uint32 n_hash[W]
    constexpr int A_NM = A_NH + BS_MAXW * 4;             // uint32 n_meta[W]
    constexpr int A_END = A_NM + BS_MAXW * 4;
    constexpr int TB_ROWS = 32;
    constexpr int A_TB_END = TB_ROWS * BS_MAXW * 4 + TB_ROWS * 2 + TB_ROWS;
    constexpr int ARENA = (A_END > A_TB_END ? A_END : A_TB_END);
    __shared__ __attribute__((aligned(16))) unsigned char arena[(ARENA + 15) / 16 * 16];
    half_t *sc_row = (half_t *)(arena + A_SC);
    float *bg_row = (float *)(arena + A_BG);
    float *c_score = (float *)(arena + A_CS);
    uint32_t *c_hash = (uint32_t *)(arena + A_CH);
    uint16_t *c_state = (uint16_t *)(arena + A_CT);
    int *tag = (int *)(arena + A_TG);
    float *n_score = (float *)(arena + A_NS);
    uint32_t *n_hash = (uint32_t *)(arena + A_NH);
    uint32_t *n_meta = (uint32_t *)(arena + A_NM);
    uint32_t *tb_tile = (uint32_t *)arena;                                   // [TB_ROWS][W]
    uint16_t *tb_state = (uint16_t *)(arena + TB_ROWS * BS_MAXW * 4);        // [TB_ROWS]
    int8_t *tb_move = (int8_t *)(arena + TB_ROWS * BS_MAXW * 4 + TB_ROWS * 2);  // [TB_ROWS]

    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    VAR_SETUP(T)
    const half_t *sn = scores + so * K;
    const float *bn = bwd + bo * S;
    uint32_t *tr = trace + bo * W;
    const uint32_t mask = S - 1;

    // ---- seed (beam_search.cpp:165-198): threshold = W-th largest back-guide at t = 0 ----
    float g0[GPL];
#pragma unroll
    for (int i = 0; i < GPL; ++i) g0[i] = bn[i * 64 + lane];
    uint32_t thr_key = 0;  // lowest
    if (W < S) {
        // radix select: largest key with count(key >= thr) >= W
        uint32_t k = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = k | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < GPL; ++i) cnt += __popcll(__ballot(f2key(g0[i]) >= cand));
            if (cnt >= W) k = cand;
        }
        thr_key = k;
    }
    int width = 0;
    {
        int base_cnt = 0;
#pragma unroll
        for (int i = 0; i < GPL; ++i) {
            const bool keep = f2key(g0[i]) >= thr_key;
            const unsigned long long bal = __ballot(keep);
            const int idx = base_cnt + lanes_below(bal, lane);
            if (keep && idx < W) {
                const uint32_t st = i * 64 + lane;
                n_hash[idx] = crc32c_bits(0x12345678u, st, 32);
                n_meta[idx] = st;
                n_score[idx] = 0.0f;
            }
            base_cnt += __popcll(bal);
        }
        width = base_cnt < W ? base_cnt : W;
    }
    __syncthreads();
    uint32_t p_hash = 0, p_state = 0;
    float p_score = 0.0f;
    if (lane < width) {
        p_hash = n_hash[lane];
        p_state = n_meta[lane] & 0xffffu;
        p_score = n_score[lane];
    }

    // prefetch row 0 of scores and row 1 of guides
    half8_t rs[RPL];
    float rg[GPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i)
        if ((i * 64 + lane) * 8 < K) rs[i] = *(const half8_t *)(sn + (i * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < GPL; ++i) rg[i] = bn[(size_t)S + i * 64 + lane];

    for (int blk = 0; blk < T; ++blk) {
        __syncthreads();  // previous block's LDS reads are complete
#pragma unroll
        for (int i = 0; i < RPL; ++i)
            if ((i * 64 + lane) * 8 < K) *(half8_t *)(sc_row + (i * 64 + lane) * 8) = rs[i];
#pragma unroll
        for (int i = 0; i < GPL; ++i) bg_row[i * 64 + lane] = rg[i];
        __syncthreads();
        if (blk + 1 < T) {
#pragma unroll
            for (int i = 0; i < RPL; ++i)
                if ((i * 64 + lane) * 8 < K)
                    rs[i] = *(const half8_t *)(sn + (size_t)(blk + 1) * K + (i * 64 + lane) * 8);
#pragma unroll
            for (int i = 0; i < GPL; ++i) rg[i] = bn[(size_t)(blk + 2) * S + i * 64 + lane];
        }
        const bool active = lane < width;
        const int w4 = width << 2;
        const int cnt = 5 * width;
        float my_max = FLT_LOWEST;
        float stay_sc = FLT_LOWEST;
        // ---- expand: steps at slot 4e+b (beam_search.cpp:236-260), stay at 4w+e (:262-271) ----
        if (active) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t ns = ((p_state << 2) & mask) | (uint32_t)b;
                const uint32_t mi = (ns << 2) + (p_state >> (BITS - 2));
                const float sc = (p_score + clampf((float)sc_row[mi], clampv)) + bg_row[ns];
                c_score[4 * lane + b] = sc;
                c_hash[HT ? (b * BS_MAXW + lane) : (4 * lane + b)] = crc32c_bits(p_hash, (uint32_t)b, 2);
                c_state[4 * lane + b] = (uint16_t)ns;
                my_max = fmaxf(my_max, sc);
            }
            stay_sc = (p_score + stay) + bg_row[p_state];
            c_score[w4 + lane] = stay_sc;
            c_hash[HT ? (4 * BS_MAXW + lane) : (w4 + lane)] = p_hash;
            c_state[w4 + lane] = (uint16_t)p_state;
            my_max = fmaxf(my_max, stay_sc);
            tag[4 * lane + 0] = -1;
            tag[4 * lane + 1] = -1;
            tag[4 * lane + 2] = -1;
            tag[4 * lane + 3] = -1;
        }
        __syncthreads();
        // ---- merge stays with equal-hash steps (beam_search.cpp:273-305).  The reference's
        //      4096-bit presence filter has no false negatives, so "compare against every step
        //      with the same newest base" is the same set of matches. ----
        uint32_t mm = 0;
        const int lb = p_state & 3;
        if (active) {
            if constexpr (HT) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 *hrow = (const u32x4 *)(c_hash + lb * BS_MAXW);
#pragma unroll
                for (int i = 0; i < BS_MAXW / 4; ++i) {
                    const u32x4 h = hrow[i];
                    mm |= (h[0] == p_hash ? 1u : 0u) << (4 * i) | (h[1] == p_hash ? 2u : 0u) << (4 * i) |
                          (h[2] == p_hash ? 4u : 0u) << (4 * i) | (h[3] == p_hash ? 8u : 0u) << (4 * i);
                }
                mm &= (width >= 32) ? 0xffffffffu : ((1u << width) - 1u);   // slots >= width hold an earlier block's hashes
            } else {
                for (int e2 = 0; e2 < width; ++e2) {
                    if (c_hash[(e2 << 2) | lb] == p_hash) mm |= (1u << e2);
                }
            }
        }
        if (__ballot(mm != 0) != 0ull) {
            const int tgt = (mm != 0) ? ((__ffs(mm) - 1) * 4 + lb) : 0;
            if (mm != 0) tag[tgt] = lane;
            __syncthreads();
            const bool clash = (mm != 0) && ((__popc(mm) > 1) || (tag[tgt] != lane));
            if (__ballot(clash) == 0ull) {
                // common case: every stay folds with at most one step and no step is shared
                if (mm != 0) {
                    const float a = stay_sc, b = c_score[tgt];
                    const float folded = dm_log_sum_exp2(a, b);
                    if (a > b) {
                        c_score[w4 + lane] = folded;
                        c_score[tgt] = FLT_LOWEST;
                    } else {
                        c_score[tgt] = folded;
                        c_score[w4 + lane] = FLT_LOWEST;
                    }
                    my_max = fmaxf(my_max, folded);
                }
            } else {
                // rare (hash collision): replay the reference's sequential order exactly
                volatile float *vs = c_score;
                for (int e = 0; e < width; ++e) {
                    uint32_t m = __shfl(mm, e, 64);
                    const int elb = __shfl(lb, e, 64);
                    while (m) {
                        const int e2 = __ffs(m) - 1;
                        m &= m - 1;
                        const int si = w4 + e, ti = (e2 << 2) | elb;
                        const float a = vs[si], b = vs[ti];
                        const float folded = dm_log_sum_exp2(a, b);
                        if (lane == 0) {
                            if (a > b) {
                                vs[si] = folded;
                                vs[ti] = FLT_LOWEST;
                            } else {
                                vs[ti] = folded;
                                vs[si] = FLT_LOWEST;
                            }
                        }
                        my_max = fmaxf(my_max, folded);
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();
        const float max_score = wave_max(my_max);

        // ---- cut-off (beam_search.cpp:310-396) ----
        float cs[3];
        bool cv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int slot = g * 64 + lane;
            cv[g] = slot < cnt;
            cs[g] = cv[g] ? c_score[slot] : FLT_LOWEST;
        }
        float cutoff = max_score - log_cut;
        auto count_ge = [&](float c) {
            int k = 0;
#pragma unroll
            for (int g = 0; g < 3; ++g) k += __popcll(__ballot(cv[g] && cs[g] >= c));
            return k;
        };
        int ec = count_ge(cutoff);
        if (ec > W) {
            const int minw = (W * 8) / 10;
            float lo = cutoff, hi = max_score;
            int guesses = 1;
            while ((ec > W || ec < minw) && guesses < 10) {
                if (ec > W) {
                    lo = cutoff;
                    cutoff = (cutoff + hi) / 2.0f;
                } else {
                    hi = cutoff;
                    cutoff = (cutoff + lo) / 2.0f;
                }
                ec = count_ge(cutoff);
                ++guesses;
            }
            if (guesses == 10) {
                cutoff = hi;
                ec = count_ge(cutoff);
            }
            if (ec > W) ec = W;
        }
        // ---- compaction in slot order, first W (beam_search.cpp:398-409) ----
        {
            int base_cnt = 0;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const bool keep = cv[g] && cs[g] >= cutoff;
                const unsigned long long bal = __ballot(keep);
                const int idx = base_cnt + lanes_below(bal, lane);
                if (keep && idx < W) {
                    const int slot = g * 64 + lane;
                    const bool is_stay = slot >= w4;
                    const uint32_t prev = is_stay ? (uint32_t)(slot - w4) : (uint32_t)(slot >> 2);
                    n_score[idx] = cs[g];
                    n_hash[idx] = c_hash[HT ? (is_stay ? (4 * BS_MAXW + (slot - w4)) : ((slot & 3) * BS_MAXW + (slot >> 2))) : slot];
                    n_meta[idx] = (uint32_t)c_state[slot] | (prev << 16) | ((is_stay ? 1u : 0u) << 24);
                }
                base_cnt += __popcll(bal);
            }
        }
        __syncthreads();
        uint32_t meta = 0;
        if (lane < ec) {
            p_score = n_score[lane];
            p_hash = n_hash[lane];
            meta = n_meta[lane];
            p_state = meta & 0xffffu;
        } else {
            p_score = FLT_LOWEST;
        }
        // ---- last block: best element to slot 0 (beam_search.cpp:413-424; strict >, first) ----
        if (blk == T - 1) {
            float best = (lane < ec) ? p_score : FLT_LOWEST;
            const float bmax = wave_max(best);
            const unsigned long long who = __ballot((lane < ec) && (p_score == bmax));
            int bi = (who != 0ull) ? (__ffsll((long long)who) - 1) : 0;
            if (!(bmax > FLT_LOWEST)) bi = 0;
            const float s0 = __shfl(p_score, 0, 64), sb = __shfl(p_score, bi, 64);
            const uint32_t h0 = __shfl(p_hash, 0, 64), hb = __shfl(p_hash, bi, 64);
            const uint32_t m0 = __shfl(meta, 0, 64), mb = __shfl(meta, bi, 64);
            if (lane == 0) {
                p_score = sb; p_hash = hb; meta = mb;
            } else if (lane == bi) {
                p_score = s0; p_hash = h0; meta = m0;
            }
            p_state = meta & 0xffffu;
        }
        // ---- store (beam_search.cpp:426-437) ----
        if (lane < ec) {
            p_score -= bg_row[p_state];
            tr[(size_t)(blk + 1) * W + lane] = meta;
        }
        width = ec;
    }
    __syncthreads();
    __threadfence_block();

    // ---- trace back (beam_search.cpp:448-455), TB_ROWS blocks at a time through LDS ----
    uint8_t ei = 0;
    for (int hi_blk = T; hi_blk >= 1; hi_blk -= TB_ROWS) {
        const int lo_blk = (hi_blk - (TB_ROWS - 1) > 1) ? (hi_blk - (TB_ROWS - 1)) : 1;  // rows lo..hi
        const int rows = hi_blk - lo_blk + 1;
        const int words = rows * W;
        // bulk copy (all loads issued before the first LDS store)
        constexpr int PER = TB_ROWS * BS_MAXW / 64;
        uint32_t tmp[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            tmp[q] = (i < words) ? tr[(size_t)lo_blk * W + i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            if (i < words) tb_tile[i] = tmp[q];
        }
        __syncthreads();
        if (lane == 0) {
            for (int r = rows - 1; r >= 0; --r) {
                const uint32_t m = tb_tile[r * W + ei];
                tb_state[r] = (uint16_t)(m & 0xffffu);
                tb_move[r] = ((m >> 24) & 1u) ? 0 : 1;
                ei = (uint8_t)((m >> 16) & 0xffu);
            }
        }
        __syncthreads();
        ei = (uint8_t)__shfl((int)ei, 0, 64);
        if (lane < rows) {
            const int blk = lo_blk + lane - 1;  // beam row b describes block b-1
            path_state[so + blk] = tb_state[lane];
            moves[so + blk] = (blk == 0) ? (int8_t)1 : tb_move[lane];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k2, 64-lane form (round 5).  The step of beam_search_kernel is bound by VALU issue, not by latency or HBM: ~410 vector
// instructions x 4 cycles = the 1600 cycles per step and SIMD that 16384 chunks x 1666 steps in 20.4 ms come to.  Its beam
// of <= 32 elements leaves half of the wave idle in the two largest parts, so here
//   * lane (e, h) = (lane & 31, lane >> 5) holds beam element e in BOTH halves; half h expands the steps with newest base
//     2h and 2h + 1 and tests its stay against 16 of the 32 candidates of the merge row; the halves exchange their 16-bit
//     masks with one v_permlane32_swap;
//   * a match bit costs v_cmp + v_addc (carry-in = the compare) instead of v_cmp + v_cndmask + shift / or;
//   * candidates live in FIXED slots (steps 4e + b, stays 128 + e) with their trace word (state | prev << 16 | stay << 24)
//     written at expansion, so the compaction is ballot + v_mbcnt + three copies (order of the kept candidates = the
//     reference's: steps in slot order, then stays);
//   * the wave maximum is six v_max_f32_dpp (wave_max above);
//   * tag[] and the new front n_*[] live inside the score row's LDS (dead between the expansion and the next block), so
//     the arena is 4992 B and 32 waves fit a CU (5888 B: 27).
// Same candidates, same matches, same order, same arithmetic: outputs are bit-identical to beam_search_kernel.
// ---------------------------------------------------------------------------------------------
// m = 2 m + (h == q) for four candidates, h3 first: the compare's lane mask is the carry-in of the add.  gfx950 wants two wait
// states between a VALU write of an SGPR pair and a VALU read of it as a mask, so the four compares go first (the trailing
// s_nop 1: the result may be read by a v_permlane32_swap, which has the same rule for VGPRs).
#define BS_MATCH4(m, h3, h2, h1, h0, q)                                                                              \
    {                                                                                                                \
        unsigned long long c0_, c1_, c2_, c3_;                                                                       \
        asm("v_cmp_eq_u32_e64 %1, %5, %9\n\t"                                                                        \
            "v_cmp_eq_u32_e64 %2, %6, %9\n\t"                                                                        \
            "v_cmp_eq_u32_e64 %3, %7, %9\n\t"                                                                        \
            "v_cmp_eq_u32_e64 %4, %8, %9\n\t"                                                                        \
            "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"                                                               \
            "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"                                                               \
            "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"                                                               \
            "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"                                                               \
            "s_nop 1"                                                                                                \
            : "+v"(m), "=&s"(c0_), "=&s"(c1_), "=&s"(c2_), "=&s"(c3_)                                                \
            : "v"(h3), "v"(h2), "v"(h1), "v"(h0), "v"(q));                                                           \
    }

template <int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void beam_search64_kernel(
        const half_t *__restrict__ scores,   // [N][T][4S]
        const float *__restrict__ bwd,       // [N][T+1][S]
        uint32_t *__restrict__ trace,        // [N][T+1][W]  state | prev<<16 | stay<<24
        uint16_t *__restrict__ path_state,   // [N][T]  full k-mer state per block
        int8_t *__restrict__ moves,          // [N][T]
        int T, int W, float log_cut, float stay, float clampv, VarIdx vi) {
    constexpr int K = 4 * S;
    constexpr int BITS = (S == 64) ? 6 : (S == 256) ? 8 : (S == 1024) ? 10 : 12;
    constexpr int RPL = (K / 64 / 8) > 0 ? (K / 64 / 8) : 1;  // half8 loads per lane per score row
    constexpr int GPL = S / 64;      // floats per lane for one guide row
    constexpr int STAY0 = 4 * BS_MAXW;                   // first stay slot
    constexpr int A_SC = 0;                              // half_t sc_row[K]
    constexpr int A_BG = A_SC + K * 2;                   // float bg_row[S]
    constexpr int A_CS = A_BG + S * 4;                   // float c_score[BS_CAND]     steps 4e + b, stays STAY0 + e
    // hash rows of the four newest bases lie HROW = 36 words apart (round 6; was 32): the merge test reads the row of the lane's own
    // base with ds_read_b128, and rows 128 B apart put bases 0 / 2 and 1 / 3 on the same bank quads (SQ_LDS_BANK_CONFLICT 0.12 of this
    // kernel's cycles, profiles/r06_j_pmc_clock_hac_q8_n16384.json); 144 B apart the four rows touch four different quads
    constexpr int HROW = BS_MAXW + 4;
    constexpr int HSTAY = 4 * HROW;                      // first stay hash
    constexpr int A_CH = A_CS + BS_CAND * 4;             // uint32 c_hash[4 * HROW + W]  steps b * HROW + e (base-major), stays HSTAY + e
    constexpr int A_CM = A_CH + (4 * HROW + BS_MAXW) * 4;   // uint32 c_meta[BS_CAND]     as c_score
    // tag[] (merge phase) and the new front n_*[] (compaction -> top of the next block) are only alive while the score row is
    // dead (it is read by the expansion alone and rewritten from registers at the top of a block, after the front has been
    // loaded), so for K >= 448 they live inside it; the accesses are LDS operations of ONE wave, which execute in order
    constexpr bool IN_ROW = K * 2 >= (4 * BS_MAXW + 3 * BS_MAXW) * 4;
    constexpr int A_TG = IN_ROW ? A_SC : A_CM + BS_CAND * 4;   // int tag[4W]
    constexpr int A_NS = A_TG + 4 * BS_MAXW * 4;         // float n_score[W]
    constexpr int A_NH = A_NS + BS_MAXW * 4;             // uint32 n_hash[W]
    constexpr int A_NM = A_NH + BS_MAXW * 4;             // uint32 n_meta[W]
    constexpr int A_END = IN_ROW ? A_CM + BS_CAND * 4 : A_NM + BS_MAXW * 4;
    constexpr int TB_ROWS = 32;
    constexpr int A_TB_END = TB_ROWS * BS_MAXW * 4 + TB_ROWS * 2 + TB_ROWS;
    constexpr int ARENA = (A_END > A_TB_END ? A_END : A_TB_END);
    __shared__ __attribute__((aligned(16))) unsigned char arena[(ARENA + 15) / 16 * 16];
    half_t *sc_row = (half_t *)(arena + A_SC);
    float *bg_row = (float *)(arena + A_BG);
    float *c_score = (float *)(arena + A_CS);
    uint32_t *c_hash = (uint32_t *)(arena + A_CH);
    uint32_t *c_meta = (uint32_t *)(arena + A_CM);
    int *tag = (int *)(arena + A_TG);
    float *n_score = (float *)(arena + A_NS);
    uint32_t *n_hash = (uint32_t *)(arena + A_NH);
    uint32_t *n_meta = (uint32_t *)(arena + A_NM);
    uint32_t *tb_tile = (uint32_t *)arena;                                   // [TB_ROWS][W]
    uint16_t *tb_state = (uint16_t *)(arena + TB_ROWS * BS_MAXW * 4);        // [TB_ROWS]
    int8_t *tb_move = (int8_t *)(arena + TB_ROWS * BS_MAXW * 4 + TB_ROWS * 2);  // [TB_ROWS]
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    const int e = lane & 31, hf = lane >> 5;
    VAR_SETUP(T)
    const half_t *sn = scores + so * K;
    const float *bn = bwd + bo * S;
    uint32_t *tr = trace + bo * W;
    const uint32_t mask = S - 1;

    // ---- seed (beam_search.cpp:165-198): threshold = W-th largest back-guide at t = 0 ----
    float g0[GPL];
#pragma unroll
    for (int i = 0; i < GPL; ++i) g0[i] = bn[i * 64 + lane];
    uint32_t thr_key = 0;  // lowest
    if (W < S) {
        // radix select: largest key with count(key >= thr) >= W
        uint32_t k = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = k | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < GPL; ++i) cnt += __popcll(__ballot(f2key(g0[i]) >= cand));
            if (cnt >= W) k = cand;
        }
        thr_key = k;
    }
    int width = 0;
    {
        int base_cnt = 0;
#pragma unroll
        for (int i = 0; i < GPL; ++i) {
            const bool keep = f2key(g0[i]) >= thr_key;
            const unsigned long long bal = __ballot(keep);
            const int idx = base_cnt + lanes_below(bal, lane);
            if (keep && idx < W) {
                const uint32_t st = i * 64 + lane;
                n_hash[idx] = crc32c_bits(0x12345678u, st, 32);
                n_meta[idx] = st;
                n_score[idx] = 0.0f;
            }
            base_cnt += __popcll(bal);
        }
        width = base_cnt < W ? base_cnt : W;
    }
    __syncthreads();
    // beam element e in both halves of the wave (slots >= width: stale values, never used unmasked)
    uint32_t p_hash = n_hash[e], p_state = n_meta[e] & 0xffffu;
    float p_score = (e < width) ? n_score[e] : 0.0f;

    // prefetch row 0 of scores and row 1 of guides
    half8_t rs[RPL];
    float rg[GPL];
    constexpr bool FULL = RPL * 64 * 8 <= K;    // every lane holds a piece of every score-row load (S >= 128)
#pragma unroll
    for (int i = 0; i < RPL; ++i)
        if (FULL || (i * 64 + lane) * 8 < K) rs[i] = *(const half8_t *)(sn + (i * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < GPL; ++i) rg[i] = bn[(size_t)S + i * 64 + lane];

    // slot -> hash index of this lane's three compaction slots (steps are base-major)
    const int hidx0 = (lane & 3) * HROW + (lane >> 2), hidx1 = hidx0 + 16;
    const uint32_t prev16 = (uint32_t)e << 16;

    for (int blk = 0; blk < T; ++blk) {
        __syncthreads();  // previous block's LDS reads are complete
#pragma unroll
        for (int i = 0; i < RPL; ++i)
            if (FULL || (i * 64 + lane) * 8 < K) *(half8_t *)(sc_row + (i * 64 + lane) * 8) = rs[i];
#pragma unroll
        for (int i = 0; i < GPL; ++i) bg_row[i * 64 + lane] = rg[i];
        __syncthreads();
        if (blk + 1 < T) {
#pragma unroll
            for (int i = 0; i < RPL; ++i)
                if (FULL || (i * 64 + lane) * 8 < K)
                    rs[i] = *(const half8_t *)(sn + (size_t)(blk + 1) * K + (i * 64 + lane) * 8);
#pragma unroll
            for (int i = 0; i < GPL; ++i) rg[i] = bn[(size_t)(blk + 2) * S + i * 64 + lane];
        }
        const bool act = e < width;
        const int w4 = width << 2;
        float my_max = FLT_LOWEST;
        float stay_sc = FLT_LOWEST;
        // ---- expand: steps at slot 4e+b (beam_search.cpp:236-260), stay at STAY0+e (:262-271); half hf takes b = 2hf, 2hf+1 ----
        if (act) {
            const uint32_t b0 = 2u * (uint32_t)hf;
            const uint32_t ns0 = ((p_state << 2) & mask) | b0;
            const uint32_t mi0 = (ns0 << 2) + (p_state >> (BITS - 2));
            const f32x2 bg = *(const f32x2 *)(bg_row + ns0);
            const float sc0 = (p_score + clampf((float)sc_row[mi0], clampv)) + bg[0];
            const float sc1 = (p_score + clampf((float)sc_row[mi0 + 4], clampv)) + bg[1];
            *(f32x2 *)(c_score + 4 * e + b0) = f32x2{sc0, sc1};
            c_hash[b0 * HROW + e] = crc32c_bits(p_hash, b0, 2);
            c_hash[(b0 + 1u) * HROW + e] = crc32c_bits(p_hash, b0 + 1u, 2);
            *(u32x2 *)(c_meta + 4 * e + b0) = u32x2{ns0 | prev16, ns0 | 1u | prev16};
            *(u32x2 *)(tag + 4 * e + b0) = u32x2{0xffffffffu, 0xffffffffu};
            my_max = fmaxf(sc0, sc1);
            if (hf == 0) {
                stay_sc = (p_score + stay) + bg_row[p_state];
                c_score[STAY0 + e] = stay_sc;
                c_hash[HSTAY + e] = p_hash;
                c_meta[STAY0 + e] = p_state | prev16 | (1u << 24);
                my_max = fmaxf(my_max, stay_sc);
            }
        }
        __syncthreads();
        // ---- merge stays with equal-hash steps (beam_search.cpp:273-305).  The reference's 4096-bit presence filter has
        //      no false negatives, so "compare against every step with the same newest base" is the same set of matches;
        //      half hf tests candidates 16 hf .. 16 hf + 15 of that row. ----
        const int lb = p_state & 3;
        uint32_t mm;
        {
            const u32x4 *hrow = (const u32x4 *)(c_hash + lb * HROW + 16 * hf);
            const u32x4 q0 = hrow[0], q1 = hrow[1], q2 = hrow[2], q3 = hrow[3];
            uint32_t m16 = 0;
            BS_MATCH4(m16, q3[3], q3[2], q3[1], q3[0], p_hash)
            BS_MATCH4(m16, q2[3], q2[2], q2[1], q2[0], p_hash)
            BS_MATCH4(m16, q1[3], q1[2], q1[1], q1[0], p_hash)
            BS_MATCH4(m16, q0[3], q0[2], q0[1], q0[0], p_hash)
            // lower half: own 16 bits | upper half's << 16
            const auto sw = __builtin_amdgcn_permlane32_swap(m16, m16, false, false);
            mm = sw[0] | (sw[1] << 16);
            mm &= (width >= 32) ? 0xffffffffu : ((1u << width) - 1u);   // slots >= width hold an earlier block's hashes
            if (!(lane < width)) mm = 0;                                 // one lane per stay does the folding
        }
        if (__ballot(mm != 0) != 0ull) {
            const int tgt = (mm != 0) ? ((__ffs(mm) - 1) * 4 + lb) : 0;
            if (mm != 0) tag[tgt] = lane;
            __syncthreads();
            const bool clash = (mm != 0) && ((__popc(mm) > 1) || (tag[tgt] != lane));
            if (__ballot(clash) == 0ull) {
                // common case: every stay folds with at most one step and no step is shared
                if (mm != 0) {
                    const float a = stay_sc, b = c_score[tgt];
                    const float folded = dm_log_sum_exp2(a, b);
                    if (a > b) {
                        c_score[STAY0 + lane] = folded;
                        c_score[tgt] = FLT_LOWEST;
                    } else {
                        c_score[tgt] = folded;
                        c_score[STAY0 + lane] = FLT_LOWEST;
                    }
                    my_max = fmaxf(my_max, folded);
                }
            } else {
                // rare (hash collision): replay the reference's sequential order exactly
                volatile float *vs = c_score;
                for (int e1 = 0; e1 < width; ++e1) {
                    uint32_t m = __shfl(mm, e1, 64);
                    const int elb = __shfl(lb, e1, 64);
                    while (m) {
                        const int e2 = __ffs(m) - 1;
                        m &= m - 1;
                        const int si = STAY0 + e1, ti = (e2 << 2) | elb;
                        const float a = vs[si], b = vs[ti];
                        const float folded = dm_log_sum_exp2(a, b);
                        if (lane == 0) {
                            if (a > b) {
                                vs[si] = folded;
                                vs[ti] = FLT_LOWEST;
                            } else {
                                vs[ti] = folded;
                                vs[si] = FLT_LOWEST;
                            }
                        }
                        my_max = fmaxf(my_max, folded);
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();
        const float max_score = wave_max(my_max);

        // ---- cut-off (beam_search.cpp:310-396): slots lane, 64 + lane (steps) and STAY0 + lane (stays).  Which of a lane's
        //      three slots hold candidates is a uniform lane mask (vm*), so a count is three v_cmp into SGPR pairs + SALU ----
        const unsigned long long vm0 = (w4 >= 64) ? ~0ull : ((1ull << w4) - 1ull);
        const unsigned long long vm1 = (w4 >= 128) ? ~0ull : ((w4 > 64) ? ((1ull << (w4 - 64)) - 1ull) : 0ull);
        const unsigned long long vm2 = (1ull << width) - 1ull;            // width <= 32
        const float cs0 = c_score[lane], cs1 = c_score[64 + lane], cs2 = c_score[STAY0 + lane];   // (unmasked slots: stale, in-arena)
        float cutoff = max_score - log_cut;
        unsigned long long k0, k1, k2;      // candidates >= cutoff
        auto count_ge = [&](float c) {
            k0 = __builtin_amdgcn_ballot_w64(cs0 >= c) & vm0;
            k1 = __builtin_amdgcn_ballot_w64(cs1 >= c) & vm1;
            k2 = __builtin_amdgcn_ballot_w64(cs2 >= c) & vm2;
            return __popcll(k0) + __popcll(k1) + __popcll(k2);
        };
        int ec = count_ge(cutoff);
        if (ec > W) {
            const int minw = (W * 8) / 10;
            float lo = cutoff, hi = max_score;
            int guesses = 1;
            while ((ec > W || ec < minw) && guesses < 10) {
                // (the reference halves towards hi or towards lo: the same sum either way round)
                if (ec > W) lo = cutoff;
                else hi = cutoff;
                cutoff = (lo + hi) / 2.0f;
                ec = count_ge(cutoff);
                ++guesses;
            }
            if (guesses == 10) {
                cutoff = hi;
                ec = count_ge(cutoff);
            }
            if (ec > W) ec = W;
        }
        // ---- compaction in slot order, first W (beam_search.cpp:398-409) ----
        {
            const int i0 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(k0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)k0, 0u));
            if (__builtin_amdgcn_inverse_ballot_w64(k0) && i0 < W) {
                n_score[i0] = cs0;
                n_hash[i0] = c_hash[hidx0];
                n_meta[i0] = c_meta[lane];
            }
            const uint32_t b1 = (uint32_t)__popcll(k0);
            const int i1 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(k1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)k1, b1));
            if (__builtin_amdgcn_inverse_ballot_w64(k1) && i1 < W) {
                n_score[i1] = cs1;
                n_hash[i1] = c_hash[hidx1];
                n_meta[i1] = c_meta[64 + lane];
            }
            const uint32_t b2 = b1 + (uint32_t)__popcll(k1);
            const int i2 = (int)__builtin_amdgcn_mbcnt_lo((uint32_t)k2, b2);     // k2 has no bit above 31
            if (__builtin_amdgcn_inverse_ballot_w64(k2) && i2 < W) {
                n_score[i2] = cs2;
                n_hash[i2] = c_hash[HSTAY + lane];
                n_meta[i2] = c_meta[STAY0 + lane];
            }
        }
        __syncthreads();
        uint32_t meta = n_meta[e];
        p_hash = n_hash[e];
        p_state = meta & 0xffffu;
        p_score = (e < ec) ? n_score[e] : FLT_LOWEST;
        // ---- last block: best element to slot 0 (beam_search.cpp:413-424; strict >, first) ----
        if (blk == T - 1) {
            float best = (lane < ec) ? p_score : FLT_LOWEST;
            const float bmax = wave_max(best);
            const unsigned long long who = __ballot((lane < ec) && (p_score == bmax));
            int bi = (who != 0ull) ? (__ffsll((long long)who) - 1) : 0;
            if (!(bmax > FLT_LOWEST)) bi = 0;
            const float s0 = __shfl(p_score, 0, 64), sb = __shfl(p_score, bi, 64);
            const uint32_t h0 = __shfl(p_hash, 0, 64), hb = __shfl(p_hash, bi, 64);
            const uint32_t m0 = __shfl(meta, 0, 64), mb = __shfl(meta, bi, 64);
            if (lane == 0) {
                p_score = sb; p_hash = hb; meta = mb;
            } else if (lane == bi) {
                p_score = s0; p_hash = h0; meta = m0;
            }
            p_state = meta & 0xffffu;
        }
        // ---- store (beam_search.cpp:426-437) ----
        if (e < ec) p_score -= bg_row[p_state];
        if (lane < ec) tr[(size_t)(blk + 1) * W + lane] = meta;
        width = ec;
    }
    __syncthreads();
    __threadfence_block();

    // ---- trace back (beam_search.cpp:448-455), TB_ROWS blocks at a time through LDS ----
    uint8_t ei = 0;
    for (int hi_blk = T; hi_blk >= 1; hi_blk -= TB_ROWS) {
        const int lo_blk = (hi_blk - (TB_ROWS - 1) > 1) ? (hi_blk - (TB_ROWS - 1)) : 1;  // rows lo..hi
        const int rows = hi_blk - lo_blk + 1;
        const int words = rows * W;
        // bulk copy (all loads issued before the first LDS store)
        constexpr int PER = TB_ROWS * BS_MAXW / 64;
        uint32_t tmp[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            tmp[q] = (i < words) ? tr[(size_t)lo_blk * W + i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = lane + 64 * q;
            if (i < words) tb_tile[i] = tmp[q];
        }
        __syncthreads();
        if (lane == 0) {
            for (int r = rows - 1; r >= 0; --r) {
                const uint32_t m = tb_tile[r * W + ei];
                tb_state[r] = (uint16_t)(m & 0xffffu);
                tb_move[r] = ((m >> 24) & 1u) ? 0 : 1;
                ei = (uint8_t)((m >> 16) & 0xffu);
            }
        }
        __syncthreads();
        ei = (uint8_t)__shfl((int)ei, 0, 64);
        if (lane < rows) {
            const int blk = lo_blk + lane - 1;  // beam row b describes block b-1
            path_state[so + blk] = tb_state[lane];
            moves[so + blk] = (blk == 0) ? (int8_t)1 : tb_move[lane];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k3: forward scan + posterior of the called path + sequence/qstring
//     (CPUDecoder.cpp:43-64,130; beam_search.cpp:459-517; beam_search.cpp:54-102)
// ---------------------------------------------------------------------------------------------
// PAIR: S/2 threads, thread j owns states j and j + S/2 (half the waves per chunk, half the wave reductions,
// plain arithmetic on the packed-f32 pipe); !PAIR: one state per thread (S = 64).
template <bool PAIR>
__global__ void posts_qual_kernel(const half_t *__restrict__ scores,      // [N][T][4S]
                                  const float *__restrict__ bwd,          // [N][T+1][S]
                                  const uint16_t *__restrict__ path_state,  // [N][T]
                                  const int8_t *__restrict__ moves,       // [N][T]
                                  int8_t *__restrict__ seq_out,           // [N][T]
                                  int8_t *__restrict__ qstr_out,          // [N][T]
                                  float *__restrict__ prob_tap,           // [N][T] or nullptr
                                  int T, int S, float stay, float clampv, float q_shift,
                                  float q_scale, VarIdx vi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *alpha = (float *)smem;                 // [2][S]
    float *red = alpha + 2 * S;                   // [2 * 32] reduction scratch
    float *prob = red + 64;                       // [T]
    uint16_t *pst = (uint16_t *)(prob + T);       // [T]
    int8_t *pmv = (int8_t *)(pst + T);            // [T]
    __shared__ int s_len;
    constexpr int NS = PAIR ? 2 : 1;

    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int NT = PAIR ? (S >> 1) : S;           // threads of this workgroup
    const int K = 4 * S;
    const int Q = S >> 2;
    const int nw = (NT + 63) >> 6;
    const int wave = tid >> 6, lane = tid & 63;
    VAR_SETUP(T)   // after the LDS carve-up above, which uses the launch-wide maximum T
    const half_t *sn = scores + so * K;
    const float *bn = bwd + bo * S;
    int sid[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) sid[k] = tid + k * NT;

    for (int i = tid; i < T; i += NT) {
        pst[i] = path_state[so + i];
        pmv[i] = moves[so + i];
    }
    // log Z = LSE_s(bwd[0][s])  (alpha[0] = 0): shift for the posterior exponent
    float logZ;
    {
        float v[NS], m = FLT_LOWEST;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            v[k] = bn[sid[k]];
            m = fmaxf(m, v[k]);
        }
        m = wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        float mm = red[0];
        for (int i = 1; i < nw; ++i) mm = fmaxf(mm, red[i]);
        __syncthreads();
        float e = 0.0f;
#pragma unroll
        for (int k = 0; k < NS; ++k) e += dm_expf(v[k] - mm);
        e = wave_sum(e);
        if (lane == 0) red[wave] = e;
        __syncthreads();
        float es = red[0];
        for (int i = 1; i < nw; ++i) es += red[i];
        logZ = mm + dm_logf(es);
        __syncthreads();
    }
    float mine[NS], bnext[NS];
    half4_t row[NS];
    int pred[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        alpha[sid[k]] = 0.0f;
        mine[k] = 0.0f;
        pred[k] = sid[k] >> 2;
        row[k] = *(const half4_t *)(sn + 4 * sid[k]);
        bnext[k] = bn[(size_t)S + sid[k]];
    }
    int p = 0;
    __syncthreads();
    constexpr float LOG2E = 1.44269504f, LN2 = 0.693147181f;
    for (int t = 0; t < T; ++t) {
        half4_t m4[NS];
        float bw[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            m4[k] = row[k];
            bw[k] = bnext[k];
        }
        if (t + 1 < T) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                row[k] = *(const half4_t *)(sn + (size_t)(t + 1) * K + 4 * sid[k]);
                bnext[k] = bn[(size_t)(t + 2) * S + sid[k]];
            }
        }
        const float *a = alpha + p * S;
        // The forward scan only feeds the posteriors (-> qstring, +-1 contract), not the called
        // bases, so it uses the hardware exp/log (v_exp_f32 / v_log_f32) instead of detmath.
        const int st = pst[t];
        float e_all = 0.0f, e_sel = 0.0f;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float v0 = mine[k] + stay, v1 = a[pred[k]] + clampf((float)m4[k][0], clampv),
                        v2 = a[pred[k] + Q] + clampf((float)m4[k][1], clampv),
                        v3 = a[pred[k] + 2 * Q] + clampf((float)m4[k][2], clampv),
                        v4 = a[pred[k] + 3 * Q] + clampf((float)m4[k][3], clampv);
            const float m = fmaxf(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)), v4);
            const float sum = __builtin_amdgcn_exp2f((v0 - m) * LOG2E) + __builtin_amdgcn_exp2f((v1 - m) * LOG2E) +
                              __builtin_amdgcn_exp2f((v2 - m) * LOG2E) + __builtin_amdgcn_exp2f((v3 - m) * LOG2E) +
                              __builtin_amdgcn_exp2f((v4 - m) * LOG2E);
            const float v = m + __builtin_amdgcn_logf(sum) * LN2;
            mine[k] = v;
            alpha[(p ^ 1) * S + sid[k]] = v;
            // posterior mass of the called k-mer and its distinct shifted neighbours
            const int s = sid[k];
            const bool in_set = (s == st) ||
                                ((s & (Q - 1)) == (st >> 2)) ||  // left-shifted:  (st >> 2) + b * Q
                                ((s >> 2) == (st & (Q - 1)));    // right-shifted: ((st << 2) % S) + b
            const float e = __builtin_amdgcn_exp2f(((v + bw[k]) - logZ) * LOG2E);
            e_all += e;
            e_sel += in_set ? e : 0.0f;
        }
        p ^= 1;
        e_all = wave_sum(e_all);
        e_sel = wave_sum(e_sel);
        if (lane == 0) {
            red[(t & 1) * 32 + wave] = e_all;
            red[(t & 1) * 32 + 16 + wave] = e_sel;
        }
        __syncthreads();  // alpha[p] and red[] visible
        // every wave folds the partials itself (broadcast LDS reads): no single thread becomes the
        // straggler that the next barrier waits for; clamp and the ^0.4 happen after the scan
        float ta = 0.0f, ts = 0.0f;
        for (int i = 0; i < nw; ++i) {
            ta += red[(t & 1) * 32 + i];
            ts += red[(t & 1) * 32 + 16 + i];
        }
        // (the IEEE division only in the wave whose thread stores it: wave-uniform branch)
        if (wave == ((t & (NT - 1)) >> 6)) {
            if (tid == (t & (NT - 1))) prob[t] = ts / ta;
        }
    }
    __syncthreads();
    for (int i = tid; i < T; i += NT) {  // beam_search.cpp:505-506
        const float pr = fminf(fmaxf(prob[i], 0.0f), 1.0f);
        prob[i] = powf(pr, 0.4f);
    }
    __syncthreads();
    // ---- sequence + qstring (beam_search.cpp:54-102).  The blocks that contribute to one base are
    //      the contiguous run [its move block, next move block): the thread that owns the move block
    //      walks its run (same summation order as the reference's sequential accumulation) and
    //      emits base and quality at position = (number of moves up to and including it) - 1. ----
    {
        const int per = (T + NT - 1) / NT;
        const int b0 = tid * per;
        int local = 0;
        for (int i = 0; i < per; ++i) {
            const int blk = b0 + i;
            if (blk < T) local += (blk == 0) ? 1 : pmv[blk];
        }
        int *scan = (int *)alpha;  // alpha is dead after the scan; NT ints
        scan[tid] = local;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {
            const int add = (tid >= o) ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        int pos = scan[tid] - local;  // exclusive prefix: bases before this thread's blocks
        if (tid == NT - 1) s_len = scan[tid];
        const char alphabet[4] = {'A', 'C', 'G', 'T'};
        for (int i = 0; i < per; ++i) {
            const int blk = b0 + i;
            if (blk >= T) break;
            const int mv = (blk == 0) ? 1 : pmv[blk];
            if (!mv) continue;
            const int base = pst[blk] & 3;
            float bsum = 0.0f, tsum = 0.0f;
            int r = blk;
            do {  // run of this base: its move block and the stays that follow
                const float pr = prob[r];
                const float wrong = (1.0f - pr) / 3.0f;
                const int rb = pst[r] & 3;
                bsum += pr;  // qual_data[r][rb] with rb == called base of block r
                float tot = tsum;
#pragma unroll
                for (int k = 0; k < 4; ++k) tot += (k == rb) ? pr : wrong;
                tsum = tot;
                ++r;
            } while (r < T && pmv[r] == 0);
            float e = 1.0f - (bsum / tsum);
            e = -10.0f * log10f(e);
            float q = e * q_scale + q_shift;
            q = fminf(fmaxf(q, 1.0f), 50.0f);
            seq_out[so + pos] = (int8_t)alphabet[base];
            qstr_out[so + pos] = (int8_t)(33.5f + q);
            pos += 1;
        }
    }
    __syncthreads();
    const int len = s_len;
    for (int i = tid; i < T; i += NT) {
        if (i >= len) {
            seq_out[so + i] = 0;
            qstr_out[so + i] = 0;
        }
        if (prob_tap != nullptr) prob_tap[so + i] = prob[i];
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" int mibc_launch_decode_var(hipStream_t st, const half_t *scores, int N, int T, int S, int W,
                                      float beam_cut, float stay, float clampv, float q_shift, float q_scale,
                                      float *bwd, uint32_t *trace, uint16_t *path_state, int8_t *out3,
                                      size_t plane_stride, float *prob_tap, const int *coff, const int *boff,
                                      const int *clen);
extern "C" int mibc_launch_decode(hipStream_t st, const half_t *scores, int N, int T, int S, int W,
                                  float beam_cut, float stay, float clampv, float q_shift,
                                  float q_scale, float *bwd, uint32_t *trace,
                                  uint16_t *path_state, int8_t *out3 /* [3][Nplane][T] */,
                                  size_t plane_stride, float *prob_tap) {
    return mibc_launch_decode_var(st, scores, N, T, S, W, beam_cut, stay, clampv, q_shift, q_scale, bwd, trace,
                                  path_state, out3, plane_stride, prob_tap, nullptr, nullptr, nullptr);
}

// N = number of chunks of the launch; with coff / boff / clen (device int32 [N]) chunk n is the step interval
// [coff[n], coff[n] + clen[n]) of the packed planes and owns back-guide rows [boff[n], boff[n] + clen[n] + 1);
// T = the largest clen (LDS sizing).
extern "C" int mibc_launch_decode_var(hipStream_t st, const half_t *scores, int N, int T, int S, int W,
                                      float beam_cut, float stay, float clampv, float q_shift, float q_scale,
                                      float *bwd, uint32_t *trace, uint16_t *path_state, int8_t *out3,
                                      size_t plane_stride, float *prob_tap, const int *coff, const int *boff,
                                      const int *clen) {
    const VarIdx vi{coff, boff, clen};
    if (W > BS_MAXW || W < 1 || (S != 64 && S != 256 && S != 1024) || T < 1) {
        return 1;
    }
    int8_t *moves = out3;
    int8_t *seq = out3 + plane_stride;
    int8_t *qstr = out3 + 2 * plane_stride;
    const float log_cut = (beam_cut > 0.0f) ? logf(beam_cut) : 3.402823466e+38f;
    const size_t smem1 = (size_t)2 * S * 4 + (size_t)2 * 4 * S * 2;
    static const int k1_pair = MIBC_ENV_INT("MIBC_K1_PAIR", 1);
    if (k1_pair && S >= 128)
        hipLaunchKernelGGL(bwd_scan2_kernel, dim3(N), dim3(S / 2), smem1, st, scores, bwd, T, S, stay, clampv, vi);
    else
        hipLaunchKernelGGL(bwd_scan_kernel, dim3(N), dim3(S), smem1, st, scores, bwd, T, S, stay, clampv, vi);
#ifdef MIBC_DEBUG_KERNELS
    // A/B: MIBC_K2_V=1 the 32-lane kernel of rounds 1-5 (=0: with its round 1-4 merge scan)
    static const int k2_v = MIBC_ENV_INT("MIBC_K2_V", 2);
#define K2_OLD(S_, HT_) hipLaunchKernelGGL((beam_search_kernel<S_, HT_>), dim3(N), dim3(64), 0, st, scores, bwd, trace, path_state, moves, T, W, log_cut, stay, clampv, vi)
    if (k2_v == 0 && S == 256) K2_OLD(256, false);
    else if (k2_v == 0 && S == 1024) K2_OLD(1024, false);
    else if (k2_v != 2 && S == 64) K2_OLD(64, true);
    else if (k2_v != 2 && S == 256) K2_OLD(256, true);
    else if (k2_v != 2) K2_OLD(1024, true);
    else
#undef K2_OLD
#endif
    switch (S) {
        case 64:
            hipLaunchKernelGGL((beam_search64_kernel<64>), dim3(N), dim3(64), 0, st, scores, bwd, trace,
                               path_state, moves, T, W, log_cut, stay, clampv, vi);
            break;
        case 256:
            hipLaunchKernelGGL((beam_search64_kernel<256>), dim3(N), dim3(64), 0, st, scores, bwd,
                               trace, path_state, moves, T, W, log_cut, stay, clampv, vi);
            break;
        default:
            hipLaunchKernelGGL((beam_search64_kernel<1024>), dim3(N), dim3(64), 0, st, scores, bwd,
                               trace, path_state, moves, T, W, log_cut, stay, clampv, vi);
            break;
    }
    const size_t smem3 = (size_t)(2 * S + 64) * 4 + (size_t)T * 4 + (size_t)T * 2 + (size_t)T + 16;
    static const int k3_pair = MIBC_ENV_INT("MIBC_K3_PAIR", 1);
    if (k3_pair && S >= 128)
        hipLaunchKernelGGL(posts_qual_kernel<true>, dim3(N), dim3(S / 2), smem3, st, scores, bwd, path_state, moves,
                           seq, qstr, prob_tap, T, S, stay, clampv, q_shift, q_scale, vi);
    else
        hipLaunchKernelGGL(posts_qual_kernel<false>, dim3(N), dim3(S), smem3, st, scores, bwd, path_state, moves,
                           seq, qstr, prob_tap, T, S, stay, clampv, q_shift, q_scale, vi);
    return 0;
}
