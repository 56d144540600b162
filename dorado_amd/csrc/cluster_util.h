// dorado_amd/csrc/cluster_util.h — device helpers shared by the CU-cluster kernels (lstm_cluster.hip): compile-time loops, explicit LDS / global address-space pointers, direct LDS DMA requests.
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


// ---- host side: one cluster kernel at a time per DEVICE.  Every member of a cluster spins on its peers, so the
// launcher sizes the grid as if it owned the device; two cluster launches from different streams (two engines on one
// GPU) could each end up partly resident and wait on each other until CL_SPIN_LIMIT.  Launches therefore chain through
// a per-device event: a cluster kernel starts only after the previous cluster kernel of the same device — whatever
// stream it ran on — has finished.  On one stream the wait is a no-op.
#include <mutex>
struct MibcClusterGate {
    std::mutex mut;
    hipEvent_t ev[MIBC_MAX_DEVICES] = {};
};
MibcClusterGate &mibc_cluster_gate();   // lstm_cluster.hip
struct MibcClusterLaunch {               // RAII around one cluster launch on stream s
    MibcClusterGate &g;
    hipStream_t s;
    int dev;
    explicit MibcClusterLaunch(hipStream_t s_) : g(mibc_cluster_gate()), s(s_), dev(mibc_cur_device()) {
        g.mut.lock();
        if (!g.ev[dev]) (void)hipEventCreateWithFlags(&g.ev[dev], hipEventDisableTiming);
        else (void)hipStreamWaitEvent(s, g.ev[dev], 0);
    }
    ~MibcClusterLaunch() {
        if (g.ev[dev]) (void)hipEventRecord(g.ev[dev], s);
        g.mut.unlock();
    }
};
