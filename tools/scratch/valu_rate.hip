// scratch: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32 / v_rcp_f32 on gfx950 (one 512-thread workgroup per CU, 2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = (f2){a[i], a[i] * 0.5f}; }
    const float c = 0.999f, d = 1e-3f;
    const f2 c2 = {c, c}, d2 = {d, d};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);
                else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], c2, d2);
                else if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.5f;       // 1 trans + 1 mul
                else a[i] = __builtin_amdgcn_rcpf(a[i]) + 1.0f;                         // 1 trans + 1 add
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> void run(const char *nm, float *d_out, int ncu, double elems_per_instr) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<ncu, 512>>>(d_out, iters / 10, 1.0f);
    hipEventRecord(e0);
    k<MODE><<<ncu, 512>>>(d_out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)ncu * 8 * iters * 32;            // wave-instructions of the measured kind
    printf("%-14s %.3f ms  %.2f wave-instr / clk / CU at 2.0 GHz  (%.2f cycles per wave-instruction per SIMD)\n", nm, ms,
           instr / ncu / (ms * 1e-3 * 2.0e9), (ms * 1e-3 * 2.0e9) / (instr / ncu / 4));
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float *d; hipMalloc(&d, (size_t)p.multiProcessorCount * 512 * 4);
    run<0>("v_fma_f32", d, p.multiProcessorCount, 1);
    run<1>("v_pk_fma_f32", d, p.multiProcessorCount, 2);
    run<2>("v_exp_f32+mul", d, p.multiProcessorCount, 1);
    run<3>("v_rcp_f32+add", d, p.multiProcessorCount, 1);
    return 0;
}
