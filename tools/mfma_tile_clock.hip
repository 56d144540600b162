// tools/mfma_tile_clock.hip — round 4: does the MFMA SHAPE decide the sustained clock of a 256 x 256-tile main loop?
// The wave tile of gemm256 / lstm_cluster is 64 x 128 outputs per K = 32 slab: 12 ds_read_b128 fragment reads and either
// 16 x v_mfma_f32_32x32x16_f16 or 32 x v_mfma_f32_16x16x32_f16 (same operand bytes, same 128 accumulator registers).
// This benchmark runs exactly that per-slab instruction mix (no DMA, no barriers) for >= 100 ms on RANDOM data, 8 waves
// per CU, and reports TFLOP/s, the shader clock and cycles per slab.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_tile_clock.hip -o tools/mfma_tile_clock.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// READS: 0 = operands stay in registers, 1 = 12 fragment reads per slab from LDS (different addresses per slab)
template <int SHAPE, int READS>
__global__ __launch_bounds__(512) void k(const half8 *in, float *out, unsigned long long *cyc, int iters) {
    extern __shared__ half8 lds[];   // 64 KiB: 4096 x 16 B
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) lds[i] = in[(size_t)blockIdx.x * 4096 + i];
    __syncthreads();
    half8 w[8], x[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = lds[(tid * 8 + i) & 4095];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = lds[(tid * 4 + i + 1000) & 4095];
    f16v acc32[8] = {};
    f4v acc16[32] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (READS) {
            const int base = (it * 64) & 4095;
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = lds[(base + i * 64 + (tid & 63)) & 4095];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = lds[(base + 2048 + i * 64 + (tid & 63)) & 4095];
        }
        if (SHAPE == 32) {
            // per k16 step: 4 weight fragments x 2 activation fragments; two k16 steps per slab
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        acc32[g * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks * 4 + g], x[ks * 2 + r], acc32[g * 2 + r], 0, 0, 0);
        } else {
            // one k32 step: 8 weight fragments x 4 activation fragments
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc16[g * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[g], x[r], acc16[g * 4 + r], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 32; ++i)
        for (int r = 0; r < 4; ++r) s += acc16[i][r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) atomicMax(cyc, t1 - t0);
}

template <int SHAPE, int READS>
static void run(const char *name, const half8 *d_in, float *d_out, unsigned long long *d_cyc, int ncu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<SHAPE, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<SHAPE, READS><<<ncu, 512, 65536>>>(d_in, d_out, d_cyc, iters / 4);   // warm-up (also brings the chip to its power state)
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    k<SHAPE, READS><<<ncu, 512, 65536>>>(d_in, d_out, d_cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)ncu * 8 * iters * 2.0 * 64 * 128 * 32;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("{\"data\": \"%s\", \"mfma\": \"%s\", \"frag_reads_per_slab\": %d, \"ms\": %.2f, \"tflops\": %.1f, \"frac_of_2.5PF\": %.3f, "
           "\"clock_ghz\": %.3f, \"cycles_per_slab_per_simd\": %.1f}\n",
           name, SHAPE == 32 ? "32x32x16_f16" : "16x16x32_f16", READS ? 12 : 0, ms, tf, tf / 2500.0, cyc / (ms * 1e-3) / 1e9,
           (double)cyc / ((double)iters) * 2.0 / 2.0);
    fflush(stdout);
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 200000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const size_t n = (size_t)ncu * 4096;
    std::vector<half8> h(n);
    half8 *d_in;
    float *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_in, n * sizeof(half8));
    hipMalloc(&d_out, (size_t)ncu * 512 * 4);
    hipMalloc(&d_cyc, 8);
    for (int data = 1; data >= 0; --data) {
        unsigned s = 12345;
        for (size_t i = 0; i < n; ++i)
            for (int e = 0; e < 8; ++e) {
                s = s * 1664525u + 1013904223u;
                h[i][e] = data ? (_Float16)(((float)((s >> 9) & 0x7fff) / 16384.0f - 1.0f) * 0.25f) : (_Float16)0.0f;
            }
        hipMemcpy(d_in, h.data(), n * sizeof(half8), hipMemcpyHostToDevice);
        const char *nm = data ? "random" : "zero";
        for (int rep = 0; rep < (data ? 2 : 1); ++rep) {   // random twice, interleaved: A/B on one box
            run<32, 0>(nm, d_in, d_out, d_cyc, ncu, iters);
            run<16, 0>(nm, d_in, d_out, d_cyc, ncu, iters);
            run<32, 1>(nm, d_in, d_out, d_cyc, ncu, iters);
            run<16, 1>(nm, d_in, d_out, d_cyc, ncu, iters);
        }
    }
    return 0;
}
