// dorado_amd/csrc/common.h — shared device/host helpers for the mibc HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

#define MIBC_WAVE 64

// Environment switches and ablation kernels exist only in the debug build (`make DEBUG_KERNELS=1`, used by
// tools/): the product library selects its kernels from the shape alone, so a stray MIBC_* variable can never
// change the arithmetic of a bit-exact path.
#include <stdlib.h>
#ifdef MIBC_DEBUG_KERNELS
#define MIBC_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define MIBC_ENV_INT(name, dflt) (dflt)
#endif
// libmibc.so is compiled with -fvisibility=hidden and exports exactly include/mibc.h (MIBC_API).  Test / timing hooks
// (kernel-vs-kernel comparisons, microbenchmarks, cycle stamps: mibc_debug_*) are compiled and exported only in the debug
// library libmibc_dbg.so (`make debug`), which tests/ and tools/ load when they need them (capi.dbg_lib()).
#define MIBC_HOOK extern "C" __attribute__((visibility("default")))

// ---- per-DEVICE launch state (host side).  Function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) and
// the CU count belong to a device, and one process drives every GPU of the node (one HipCaller per device on its own
// thread, api/runner_creation.cpp:85-124): both are looked up by the launching thread's CURRENT device, lock-free
// (setting the same attribute twice is idempotent; the bit only says "done on this device").
#include <atomic>
#define MIBC_MAX_DEVICES 64
static inline int mibc_cur_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (dev >= 0 && dev < MIBC_MAX_DEVICES) ? dev : 0;
}
static inline int mibc_ncu() {   // CU count of the current device
    static std::atomic<int> cus[MIBC_MAX_DEVICES];
    const int dev = mibc_cur_device();
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
#define MIBC_LDS_ATTR_ONCE(kern, bytes)                                                                       \
    do {                                                                                                      \
        static std::atomic<unsigned long long> done_{0};                                                      \
        const unsigned long long bit_ = 1ull << mibc_cur_device();                                            \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                                                \
            (void)hipFuncSetAttribute((const void *)(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                      (int)(bytes));                                                          \
            done_.fetch_or(bit_, std::memory_order_release);                                                  \
        }                                                                                                     \
    } while (0)

// D[row][col] layout of v_mfma_f32_32x32x16_f16 (cdna_hip_programming.md §3):
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),  reg in [0,16)
// A operand: lane holds A[i = lane & 31][k = 8 * (lane >> 5) + 0..7]; B likewise with j.
__device__ __forceinline__ float16_t mfma32x32x16(half8_t a, half8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float fast_sigmoid(float x) {
    // 1 / (1 + e^-x); v_exp_f32 / v_rcp_f32 (~1 ulp), saturates cleanly for large |x|
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 2 sigmoid(2x) - 1
    return fmaf(2.0f, fast_sigmoid(2.0f * x), -1.0f);
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 2) {
        return fast_tanh(v);
    }
    float s = v * fast_sigmoid(v);
    if (act == 1) {
        s = fminf(s, 3.5f);
    }
    return s;
}
