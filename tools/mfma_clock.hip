// tools/mfma_clock.hip — MFMA-only microbenchmark: what clock and what fraction of the 2.5 PF dense
// f16 peak does gfx950 sustain on nothing but MFMAs, with ZERO vs RANDOM operands, with and without LDS fragment reads
// beside them?  One 512-thread workgroup per CU (2 waves per SIMD), every wave issues ITERS x 16 independent MFMAs.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_clock.hip -o tools/mfma_clock.bin ; run: tools/mfma_clock.bin
// Prints one JSON line per variant: TFLOP/s, effective shader clock (s_memtime cycles of the longest wave / event time),
// fraction of 2.5 PF.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE, int LDS>
__global__ __launch_bounds__(512) void k(const half8 *in, float *out, unsigned long long *cyc, int iters) {
    __shared__ half8 lds[512 * 4];
    const int tid = threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(blockIdx.x * 512 + tid) * 8 + i];
        b[i] = in[(blockIdx.x * 512 + tid) * 8 + 4 + i];
        lds[tid * 4 + i] = a[i];
    }
    __syncthreads();
    f16v acc32[4] = {};
    f4v acc16[16] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {   // 4 x ds_read_b128 per 16 MFMAs: the fragment traffic of a 64 x 128 wave tile
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds[((tid + it * 7) & 511) * 4 + i];
        }
        if (SHAPE == 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[i], acc32[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc16[j * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[i], acc16[j * 4 + i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) s += acc16[i][r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) atomicMax(cyc, t1 - t0);
}

template <int SHAPE, int LDS>
static void run(const char *name, const half8 *d_in, float *d_out, unsigned long long *d_cyc, int ncu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<SHAPE, LDS><<<ncu, 512>>>(d_in, d_out, d_cyc, iters / 10);   // warm-up
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    k<SHAPE, LDS><<<ncu, 512>>>(d_in, d_out, d_cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)ncu * 8 * iters * 16 * (SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("{\"variant\": \"%s\", \"mfma\": \"%s\", \"lds_reads\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_2.5PF\": %.3f, "
           "\"wave_cycles\": %llu, \"clock_ghz\": %.3f, \"cycles_per_mfma_per_simd\": %.2f}\n",
           name, SHAPE == 32 ? "32x32x16_f16" : "16x16x32_f16", LDS, ms, tf, tf / 2500.0, cyc, cyc / (ms * 1e-3) / 1e9,
           (double)cyc / ((double)iters * 16 * 2));
    fflush(stdout);
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const size_t n = (size_t)ncu * 512 * 8;
    std::vector<half8> h(n);
    half8 *d_in;
    float *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_in, n * sizeof(half8));
    hipMalloc(&d_out, (size_t)ncu * 512 * 4);
    hipMalloc(&d_cyc, 8);
    for (int data = 0; data < 2; ++data) {
        unsigned s = 12345;
        for (size_t i = 0; i < n; ++i)
            for (int e = 0; e < 8; ++e) {
                s = s * 1664525u + 1013904223u;
                h[i][e] = data ? (_Float16)(((float)((s >> 9) & 0x7fff) / 16384.0f - 1.0f) * 0.25f) : (_Float16)0.0f;
            }
        hipMemcpy(d_in, h.data(), n * sizeof(half8), hipMemcpyHostToDevice);
        const char *nm = data ? "random" : "zero";
        run<32, 0>(nm, d_in, d_out, d_cyc, ncu, iters);
        run<32, 1>(nm, d_in, d_out, d_cyc, ncu, iters);
        run<16, 0>(nm, d_in, d_out, d_cyc, ncu, iters);
        run<16, 1>(nm, d_in, d_out, d_cyc, ncu, iters);
    }
    return 0;
}
