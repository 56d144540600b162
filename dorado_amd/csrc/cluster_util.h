// dorado_amd/csrc/cluster_util.h — device helpers shared by the CU-cluster LSTM kernels (lstm_cluster.hip,
// lstm_ws.hip): compile-time loops, explicit LDS / global address-space pointers, direct LDS DMA requests.
#pragma once
#include "common.h"

#include <utility>

typedef __attribute__((address_space(1))) unsigned int gu32;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) — a guaranteed full unroll
template <typename F, int... I>
__device__ __forceinline__ void cl_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void cl_static_for(F &&f) {
    cl_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS destinations are passed as byte addresses inside the workgroup's LDS allocation (wave-uniform)
typedef __attribute__((address_space(3))) void *lds_vptr;
#define LDSP(T) __attribute__((address_space(3))) T *
typedef const __attribute__((address_space(1))) half_t *ghalf_p;   // global-address-space pointer (no generic selects)
__device__ __forceinline__ void cl_dma16(ghalf_p g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (lds_vptr)(size_t)lds_addr, 16, 0, 0);
}
__device__ __forceinline__ void cl_dma16_sc1(ghalf_p g, unsigned lds_addr) {   // agent-scope (L1 bypass) load
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (lds_vptr)(size_t)lds_addr, 16, 0, 16);
}
__device__ __forceinline__ void cl_dma16_nt(ghalf_p g, unsigned lds_addr) {   // non-temporal (streamed once)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (lds_vptr)(size_t)lds_addr, 16, 0, 2);
}
__device__ __forceinline__ void cl_dma4_sc1(const unsigned *g, unsigned lds_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (lds_vptr)(size_t)lds_addr, 4, 0, 16);
}

