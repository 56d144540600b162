#include <hip/hip_runtime.h>
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4v mfma16x16x32(half8_t a, half8_t b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4v mfma_aw(half8_t wa, half8_t b, float4v c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(wa), "v"(b));
    return c;
}
__device__ __forceinline__ float4v mfma_vw(half8_t wv, half8_t b, float4v c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(wv), "v"(b));
    return c;
}
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * fsig(2.0f * x) - 1.0f; }

constexpr int KS = 24;
// one wave = 16 hidden units (4 gates), weights stationary in registers; loops over row tiles in LDS
__global__ __launch_bounds__(256, 1) void ws_probe(const half_t *__restrict__ W, const half_t *__restrict__ X,
                                                   half_t *__restrict__ H, float *__restrict__ cst, int T, int R) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];   // [2][KS][16][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    half8_t w[KS][4];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) w[k][g] = *(const half8_t *)(W + ((((size_t)wave * KS + k) * 4 + g) * 64 + lane) * 8);
    for (int t = 0; t < T; ++t) {
        for (int r = 0; r < R; ++r) {
            // stage tile (plain loads for the probe)
            half_t *buf = lds + (r & 1) * KS * 512;
            for (int i = tid; i < KS * 64; i += 256) *(half8_t *)(buf + i * 8) = *(const half8_t *)(X + ((size_t)(t * R + r) * KS * 64 + i) * 8);
            __syncthreads();
            float4v acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = (float4v)(0.0f);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const half8_t b = *(const half8_t *)(buf + k * 512 + l15 * 32 + ((lq ^ ((l15 >> 1) & 3)) * 8));
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = (k < 16) ? mfma_aw(w[k][g], b, acc[g]) : mfma_vw(w[k][g], b, acc[g]);
            }
            asm volatile("s_nop 15\ns_nop 3" ::: "memory");
            float4v c = *(float4v *)(cst + ((size_t)(r * 4 + wave) * 64 + lane) * 4);
            half4_t hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ig = fsig(acc[0][i]), fg = fsig(acc[1][i]), gg = ftanh(acc[2][i]), og = fsig(acc[3][i]);
                c[i] = fmaf(fg, c[i], ig * gg);
                hv[i] = (half_t)(og * ftanh(c[i]));
            }
            *(float4v *)(cst + ((size_t)(r * 4 + wave) * 64 + lane) * 4) = c;
            *(half4_t *)(H + ((size_t)(t * R + r) * 16 + l15) * 64 + wave * 16 + 4 * lq) = hv;
        }
    }
}
