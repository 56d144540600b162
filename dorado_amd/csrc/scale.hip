// dorado_amd/csrc/scale.hip — signal scaling in front of the path: the device
// side of the reference's ScalerNode (read_pipeline/nodes/ScalerNode.cpp:144-269).
//
//   read_stats_kernel     per-read shift / scale of the QUANTILE and MED_MAD strategies
//                         (ScalerNode.cpp:32-52; utils::quantile_counting, torch_utils/
//                         tensor_utils.cpp:217-245): one workgroup per read, counting histogram of
//                         the int16 samples in LDS (global scratch when the value range exceeds
//                         the LDS bins), rank search by a block scan.  Integer work: bit-exact.
//   scale_i16_f16_kernel  x -> f16((float(x) - shift) / scale)   (utils::shift_scale_tensor_i16_
//                         to_f16_inplace, tensor_utils.cpp:89-142; bit-exact contract of
//                         tests/TensorUtilsTest.cpp:121-139).  HBM-bound: 2 B in, 2 B out.
// The same affine map is also fused into the first convolution's input read (conv.hip / tx.hip,
// `ss` argument), which is how the hot path consumes raw int16 chunks without a separate pass.
// Compile with -ffp-contract=off: shift / scale are fixed IEEE operation sequences.
#include "common.h"

#define RS_THREADS 1024
#define RS_LDS_BINS 16384   // two histograms of 16384 u32 = 128 KiB

// smallest bin i (0 <= i < nb) whose inclusive prefix sum exceeds `rank`; every thread gets it.
__device__ int rs_find_rank(const uint32_t *hist, int nb, uint32_t rank, uint32_t *tsum, int *result) {
    const int tid = threadIdx.x;
    const int per = (nb + RS_THREADS - 1) / RS_THREADS;
    const int b0 = tid * per;
    uint32_t local = 0;
    for (int i = 0; i < per; ++i) {
        const int b = b0 + i;
        if (b < nb) local += hist[b];
    }
    tsum[tid] = local;
    __syncthreads();
    if (tid < 64) {  // first wave: exclusive scan of 1024 partial sums, 16 per lane
        uint32_t s = 0;
        for (int i = 0; i < 16; ++i) s += tsum[tid * 16 + i];
        uint32_t incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (tid >= o) incl += up;
        }
        uint32_t run = incl - s;
        for (int i = 0; i < 16; ++i) {
            const uint32_t v = tsum[tid * 16 + i];
            tsum[tid * 16 + i] = run;
            run += v;
        }
    }
    __syncthreads();
    const uint32_t excl = tsum[tid];
    if (local != 0 && excl <= rank && rank < excl + local) {
        uint32_t acc = excl;
        for (int i = 0; i < per; ++i) {
            const int b = b0 + i;
            acc += hist[b];
            if (acc > rank) {
                *result = b;
                break;
            }
        }
    }
    __syncthreads();
    const int r = *result;
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(RS_THREADS) void read_stats_kernel(
        const int16_t *__restrict__ sig, const long long *__restrict__ off, int strategy, float qa,
        float qb, float shift_mult, float scale_mult,
        float *__restrict__ out_ss,    // [n_reads][2] shift, scale
        float *__restrict__ out_raw,   // optional [n_reads][2]: (q_a, q_b) or (median, |dev| median)
        uint32_t *__restrict__ scratch /* [gridDim.x][2][65536] for wide-range reads */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];  // [2][RS_LDS_BINS]
    __shared__ uint32_t tsum[RS_THREADS];
    __shared__ int s_lo, s_hi, s_res;
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    const long long n = off[r + 1] - off[r];
    const int16_t *x = sig + off[r];
    if (n <= 0) {  // the reference would throw on an empty read; identity map here
        if (tid == 0) {
            out_ss[2 * r] = 0.0f;
            out_ss[2 * r + 1] = 1.0f;
            if (out_raw) out_raw[2 * r] = out_raw[2 * r + 1] = 0.0f;
        }
        return;
    }
    // ---- value range ----
    int lo = 32767, hi = -32768;
    for (long long i = tid; i < n; i += RS_THREADS) {
        const int v = x[i];
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if (tid == 0) {
        s_lo = 32767;
        s_hi = -32768;
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        atomicMin(&s_lo, lo);
        atomicMax(&s_hi, hi);
    }
    __syncthreads();
    lo = s_lo;
    hi = s_hi;
    const bool wide = (hi - lo + 1) > RS_LDS_BINS;
    // narrow: bins [lo, hi] in LDS.  wide: bins over all of int16 in global scratch (base -32768).
    uint32_t *h1 = wide ? (scratch + (size_t)r * 2 * 65536) : lds_hist;
    uint32_t *h2 = wide ? (h1 + 65536) : (lds_hist + RS_LDS_BINS);
    const int base = wide ? -32768 : lo;
    const int nb = wide ? 65536 : (hi - lo + 1);
    for (int i = tid; i < nb; i += RS_THREADS) {
        h1[i] = 0;
        h2[i] = 0;
    }
    __syncthreads();
    for (long long i = tid; i < n; i += RS_THREADS) atomicAdd(&h1[(int)x[i] - base], 1u);
    __syncthreads();
    if (wide) __threadfence();

    float ra, rb, shift, scale;
    if (strategy == 0) {
        // tensor_utils.cpp:233-242: threshold = int(q * (size - 1)), first value with count > threshold
        const uint32_t ta = (uint32_t)(int)(qa * (float)(unsigned long long)(n - 1));
        const uint32_t tb = (uint32_t)(int)(qb * (float)(unsigned long long)(n - 1));
        ra = (float)(rs_find_rank(h1, nb, ta, tsum, &s_res) + base);
        rb = (float)(rs_find_rank(h1, nb, tb, tsum, &s_res) + base);
        // ScalerNode.cpp:48-50
        shift = fmaxf(10.0f, shift_mult * (ra + rb));
        scale = fmaxf(1.0f, scale_mult * (rb - ra));
    } else {
        // ScalerNode.cpp:32-40: lower median; |x - med| in int16 arithmetic; lower median again
        const uint32_t k = (uint32_t)((n - 1) / 2);
        const int med = rs_find_rank(h1, nb, k, tsum, &s_res) + base;
        for (int i = tid; i < nb; i += RS_THREADS) {
            const uint32_t c = h1[i];
            if (c != 0) {
                const int16_t diff = (int16_t)((i + base) - med);          // wraps like the int16 tensor op
                const int16_t ad = (int16_t)(diff < 0 ? -diff : diff);     // abs(-32768) stays -32768
                // narrow: 0 <= ad < nb.  wide: index over all of int16.
                atomicAdd(&h2[wide ? ((int)ad + 32768) : (int)ad], c);
            }
        }
        __syncthreads();
        if (wide) __threadfence();
        const int madi = rs_find_rank(h2, nb, k, tsum, &s_res) + (wide ? -32768 : 0);
        ra = (float)med;
        rb = (float)madi;
        shift = ra;
        scale = rb * 1.4826f + 1e-9f;
    }
    if (tid == 0) {
        out_ss[2 * r] = shift;
        out_ss[2 * r + 1] = scale;
        if (out_raw) {
            out_raw[2 * r] = ra;
            out_raw[2 * r + 1] = rb;
        }
    }
}

// One workgroup column per read (blockIdx.y), 8 samples per thread per iteration.
__global__ __launch_bounds__(256) void scale_i16_f16_kernel(const int16_t *__restrict__ sig,
                                                            const long long *__restrict__ off,
                                                            const float *__restrict__ ss,
                                                            half_t *__restrict__ out) {
    const int r = blockIdx.y;
    const long long o0 = off[r], n = off[r + 1] - o0;
    const float shift = ss[2 * r], scale = ss[2 * r + 1];
    const int16_t *x = sig + o0;
    half_t *y = out + o0;
    // vector body on the 16-byte aligned middle, scalar head/tail
    const long long head = min(n, (long long)((8 - (o0 & 7)) & 7));
    const long long body = (n - head) / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < body; i += (long long)gridDim.x * 256) {
        typedef short short8 __attribute__((ext_vector_type(8)));
        const short8 v = *(const short8 *)(x + head + i * 8);
        half8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (half_t)(((float)v[e] - shift) / scale);
        *(half8_t *)(y + head + i * 8) = h;
    }
    if (blockIdx.x == 0) {
        for (long long i = threadIdx.x; i < head; i += 256) y[i] = (half_t)(((float)x[i] - shift) / scale);
        const long long t0 = head + body * 8;
        for (long long i = t0 + threadIdx.x; i < n; i += 256) y[i] = (half_t)(((float)x[i] - shift) / scale);
    }
}

extern "C" int mibc_launch_read_stats(hipStream_t s, const int16_t *sig, const long long *off, int n_reads,
                                      int strategy, float qa, float qb, float shift_mult, float scale_mult,
                                      float *out_ss, float *out_raw, uint32_t *scratch) {
    MIBC_LDS_ATTR_ONCE(read_stats_kernel, 2 * RS_LDS_BINS * 4);
    hipLaunchKernelGGL(read_stats_kernel, dim3(n_reads), dim3(RS_THREADS), 2 * RS_LDS_BINS * 4, s, sig, off,
                       strategy, qa, qb, shift_mult, scale_mult, out_ss, out_raw, scratch);
    return 0;
}

extern "C" int mibc_launch_scale_reads(hipStream_t s, const int16_t *sig, const long long *off, int n_reads,
                                       const float *ss, half_t *out, int blocks_per_read) {
    hipLaunchKernelGGL(scale_i16_f16_kernel, dim3(blocks_per_read, n_reads), dim3(256), 0, s, sig, off, ss, out);
    return 0;
}
