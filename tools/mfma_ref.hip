// tools/mfma_ref.hip -> tools/libmfma_ref.so — the box's own matrix-pipe reference rate, measured by bench.py in the SAME process
// right before its timed region: nothing but MFMAs on RANDOM operands (the data-dependent power draw of the
// matrix pipe sets the clock), one 512-thread workgroup per CU (2 waves per SIMD, the occupancy of the LSTM kernels), run for
// about a second so that the clock has settled where the power limit puts it.  bench.py reports roofline.bare_mfma_tf (this
// rate for the MFMA shape of the dominant kernel), roofline.bare_clock_ghz and roofline.frac_of_bare = achieved / bare: a figure
// that does not move with the box's power / clock behaviour, unlike the fraction of the 2.5 PF nominal peak.
// MEASUREMENT TOOLING (like tools/mfma_clock.hip, its command-line sibling): not part of libmibc.so, never on the product path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

// SHAPE 0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_32x32x16_f16   2: v_mfma_i32_16x16x64_i8   3: v_mfma_i32_32x32x32_i8
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_ref_kernel(const half8 *in, float *out, unsigned long long *cyc, int iters) {
    const int tid = threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(blockIdx.x * 512 + tid) * 8 + i];
        b[i] = in[(blockIdx.x * 512 + tid) * 8 + 4 + i];
    }
    f16v acc32[4] = {};
    f4v acc16[16] = {};
    i4v acci[16] = {};
    i16v acci32[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (SHAPE == 1) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[i], acc32[i], 0, 0, 0);
                else if (SHAPE == 0) acc16[j * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[i], acc16[j * 4 + i], 0, 0, 0);
                else if (SHAPE == 2) acci[j * 4 + i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i4v, a[j]), __builtin_bit_cast(i4v, b[i]), acci[j * 4 + i], 0, 0, 0);
                else acci32[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i4v, a[j]), __builtin_bit_cast(i4v, b[i]), acci32[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc32[i][r] + (float)acci32[i][r];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) s += acc16[i][r] + (float)acci[i][r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) atomicMax(cyc, t1 - t0);
}

// Runs launches of `iters` x 16 MFMAs per wave back to back until `seconds` have passed; reports the mean rate of the second
// half of the launches (the settled clock).  Returns 0, or a negative hipError.
extern "C" __attribute__((visibility("default"))) int mfma_ref_rate(int shape, double seconds, double *tflops, double *clock_ghz,
                                                                    int *launches) {
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -(int)e_; } while (0)
    if (shape < 0 || shape > 3 || !tflops) return -1;
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, dev));
    const int ncu = p.multiProcessorCount;
    const size_t n = (size_t)ncu * 512 * 8;
    std::vector<half8> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i)
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            h[i][e] = (_Float16)(((float)((s >> 9) & 0x7fff) / 16384.0f - 1.0f) * 0.25f);   // (int8: the same random bytes)
        }
    half8 *d_in = nullptr;
    float *d_out = nullptr;
    unsigned long long *d_cyc = nullptr;
    CK(hipMalloc(&d_in, n * sizeof(half8)));
    CK(hipMalloc(&d_out, (size_t)ncu * 512 * 4));
    CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_in, h.data(), n * sizeof(half8), hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 100000;
    const double flop_per_launch = (double)ncu * 8 * iters * 16 * (shape == 1 ? 2.0 * 32 * 32 * 16 : shape == 0 ? 2.0 * 16 * 16 * 32 : shape == 2 ? 2.0 * 16 * 16 * 64 : 2.0 * 32 * 32 * 32);
    auto launch = [&](int it) {
        if (shape == 0) hipLaunchKernelGGL(mfma_ref_kernel<0>, dim3(ncu), dim3(512), 0, st, d_in, d_out, d_cyc, it);
        else if (shape == 1) hipLaunchKernelGGL(mfma_ref_kernel<1>, dim3(ncu), dim3(512), 0, st, d_in, d_out, d_cyc, it);
        else if (shape == 2) hipLaunchKernelGGL(mfma_ref_kernel<2>, dim3(ncu), dim3(512), 0, st, d_in, d_out, d_cyc, it);
        else hipLaunchKernelGGL(mfma_ref_kernel<3>, dim3(ncu), dim3(512), 0, st, d_in, d_out, d_cyc, it);
    };
    launch(iters / 10);
    CK(hipStreamSynchronize(st));
    std::vector<float> ms;
    std::vector<unsigned long long> cy;
    double total = 0;
    while (total < seconds * 1e3 && ms.size() < 400) {
        CK(hipMemsetAsync(d_cyc, 0, 8, st));
        CK(hipEventRecord(e0, st));
        launch(iters);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        unsigned long long c = 0;
        CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
        ms.push_back(t);
        cy.push_back(c);
        total += t;
    }
    double tsum = 0, csum = 0;
    const size_t half = ms.size() / 2;
    for (size_t i = half; i < ms.size(); ++i) {
        tsum += ms[i];
        csum += (double)cy[i];
    }
    const size_t cnt = ms.size() - half;
    *tflops = flop_per_launch * cnt / (tsum * 1e-3) / 1e12;
    if (clock_ghz) *clock_ghz = csum / (tsum * 1e-3) / 1e9;
    if (launches) *launches = (int)ms.size();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_cyc);
    return 0;
#undef CK
}
