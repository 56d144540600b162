// tools/store_rate.hip — round 4: what does the vector store path of a CU sustain?  One 512-thread workgroup per CU (grid G),
// every wave issues `iters` 16-byte-per-lane stores (1 KB per instruction) in one of three shapes:
//   0: 1 KB contiguous;  1: 8 rows x 128 B, rows `pitch` bytes apart (the GEMM epilogues);  2: 16 rows x 64 B.
// Prints GB/s total and bytes per clock per CU for G = all CUs and G = 32 (one per XCD slot group), i.e. whether a store burst
// is bound per CU or by the memory system.   build: hipcc -O3 --offload-arch=gfx950 tools/store_rate.hip -o tools/store_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(char *out, size_t per_wg, int iters, long pitch, unsigned long long *cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char *base = out + (size_t)blockIdx.x * per_wg;
    const u4 v = {(unsigned)tid, 1u, 2u, 3u};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        size_t off;
        const size_t blk = (size_t)(it * 8 + wave);            // 1 KB block index of this instruction
        if (SHAPE == 0) off = blk * 1024 + lane * 16;
        else if (SHAPE == 1) off = ((blk >> 3) * 64 + (blk & 7) * 8 + (lane >> 3)) * (size_t)pitch % per_wg / 128 * 128 + (lane & 7) * 16;
        else off = ((blk >> 3) * 128 + (blk & 7) * 16 + (lane >> 2)) * (size_t)pitch % per_wg / 128 * 128 + ((blk >> 6) & 1) * 64 + (lane & 3) * 16;
        *(u4 *)(base + (off % per_wg)) = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) atomicMax(cyc, t1 - t0);
}
template <int SHAPE>
static void run(const char *nm, char *d, size_t per_wg, int G, int iters, long pitch, unsigned long long *dc) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<SHAPE><<<G, 512>>>(d, per_wg, iters / 8, pitch, dc);
    hipMemset(dc, 0, 8);
    hipEventRecord(a);
    k<SHAPE><<<G, 512>>>(d, per_wg, iters, pitch, dc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long c = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)G * 8 * iters * 1024.0;
    printf("{\"shape\": \"%s\", \"workgroups\": %d, \"GBps\": %.0f, \"GBps_per_cu\": %.1f, \"bytes_per_clk_per_cu\": %.1f, \"cycles_per_store_instr_per_cu\": %.1f, \"ms\": %.2f}\n",
           nm, G, bytes / ms / 1e6, bytes / ms / 1e6 / G, (double)8 * iters * 1024.0 / (double)c, (double)c / (8.0 * iters), ms);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const size_t per_wg = (size_t)256 << 20;      // 256 MB per workgroup: far beyond any cache
    char *d;
    unsigned long long *dc;
    if (hipMalloc(&d, per_wg * ncu) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&dc, 8);
    const int iters = 16384;                      // 128 MB per workgroup
    for (int G : {ncu, 32}) {
        run<0>("1 KB contiguous", d, per_wg, G, iters, 0, dc);
        for (long pitch : {256L, 512L, 1024L, 2048L, 3072L, 4096L, 8192L, 8192L + 256, 8192L + 1024, 65536L, 65536L + 256}) {
            char nm[64];
            snprintf(nm, sizeof nm, "8 rows x 128 B, pitch %ld", pitch);
            run<1>(nm, d, per_wg, G, iters, pitch, dc);
        }
        run<2>("16 rows x 64 B, pitch 8192", d, per_wg, G, iters, 8192, dc);
        run<2>("16 rows x 64 B, pitch 8448", d, per_wg, G, iters, 8448, dc);
    }
    return 0;
}
