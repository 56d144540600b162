// dorado_amd/csrc/vbz.hip — POD5 "VBZ" signal decode, StreamVByte-16 stage.
//
// Replaces the svb16 + zig-zag + delta half of pod5_get_read_complete_signal (called from
// dorado/data_loader/DataLoader.cpp:163-170; pod5-file-format 0.3.36, not vendored): after the
// zstd stage (host, libzstd) a signal row is [(n + 7) / 8 control bytes][data bytes], one control
// BIT per value (LSB first; 0 = 1 data byte, 1 = 2 data bytes, little endian); value -> zig-zag
// decode -> running sum mod 2^16 = int16 sample.
//
// One workgroup of 256 threads per signal row, tiles of 4096 values.  A thread owns 16 consecutive
// values = two control bytes, so its data length is 16 + popcount(keys) and both sequential
// dependencies (byte offsets, the delta sum) reduce to block-wide exclusive scans of one number per
// thread.  Integer work, bit-exact.  HBM: reads 1.1-2.1 B, writes 2 B per sample.
#include "common.h"

#define VBZ_THREADS 256
#define VBZ_PER 16

__device__ __forceinline__ uint32_t vbz_block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VBZ_THREADS / 64; ++w) {
        const uint32_t s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(VBZ_THREADS) void svb16_decode_kernel(
        const uint8_t *__restrict__ streams,        // concatenated svb16 streams
        const long long *__restrict__ stream_off,   // [n_rows + 1] byte offsets
        const long long *__restrict__ sample_off,   // [n_rows + 1] sample offsets
        int16_t *__restrict__ out,                  // concatenated samples
        int *__restrict__ status) {                 // [n_rows]: 0 ok, 1 stream length mismatch
    __shared__ uint32_t wsum[VBZ_THREADS / 64];
    const int row = blockIdx.x;
    const uint8_t *st = streams + stream_off[row];
    const long long slen = stream_off[row + 1] - stream_off[row];
    const long long n = sample_off[row + 1] - sample_off[row];
    int16_t *o = out + sample_off[row];
    const long long nk = (n + 7) / 8;
    long long data_pos = nk;      // running byte position of this tile's data
    uint32_t prev = 0;            // running sample value (mod 2^16)
    bool bad = nk > slen;
    for (long long t0 = 0; t0 < n && !bad; t0 += (long long)VBZ_THREADS * VBZ_PER) {
        const long long i0 = t0 + (long long)threadIdx.x * VBZ_PER;   // first value of this thread
        int cnt = 0;
        uint32_t keys = 0;
        if (i0 < n) {
            cnt = (int)((n - i0 < VBZ_PER) ? (n - i0) : VBZ_PER);
            keys = st[i0 >> 3];
            if (cnt > 8) keys |= (uint32_t)st[(i0 >> 3) + 1] << 8;
            keys &= (cnt >= 16) ? 0xffffu : ((1u << cnt) - 1u);
        }
        const uint32_t bytes = (uint32_t)cnt + (uint32_t)__popc(keys);
        uint32_t tile_bytes;
        const uint32_t boff = vbz_block_excl_scan(bytes, wsum, &tile_bytes);
        if (data_pos + (long long)tile_bytes > slen) {   // uniform: every thread sees the same totals
            bad = true;
            break;
        }
        // decode this thread's values: zig-zag, local running sum
        const uint8_t *d = st + data_pos + boff;
        uint16_t vals[VBZ_PER];
        uint32_t run = 0;
        int p = 0;
#pragma unroll
        for (int j = 0; j < VBZ_PER; ++j) {
            uint32_t u = 0;
            if (j < cnt) {
                u = d[p];
                if ((keys >> j) & 1u) u |= (uint32_t)d[p + 1] << 8;
                p += 1 + ((keys >> j) & 1u);
            }
            const uint32_t dz = (u >> 1) ^ (0u - (u & 1u));   // zig-zag decode (low 16 bits matter)
            run += dz;
            vals[j] = (uint16_t)run;
        }
        uint32_t tile_sum;
        const uint32_t soff = vbz_block_excl_scan(run & 0xffffu, wsum, &tile_sum);
        const uint32_t base = prev + soff;
        if (cnt == VBZ_PER && ((reinterpret_cast<uintptr_t>(o + i0) & 15) == 0)) {
            typedef short short8 __attribute__((ext_vector_type(8)));
            short8 a, b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = (short)(uint16_t)(base + vals[j]);
                b[j] = (short)(uint16_t)(base + vals[8 + j]);
            }
            *(short8 *)(o + i0) = a;
            *(short8 *)(o + i0 + 8) = b;
        } else {
            for (int j = 0; j < cnt; ++j) o[i0 + j] = (int16_t)(uint16_t)(base + vals[j]);
        }
        prev = (prev + tile_sum) & 0xffffu;
        data_pos += tile_bytes;
    }
    if (threadIdx.x == 0) status[row] = (bad || data_pos != slen) ? 1 : 0;
}

extern "C" int mibc_launch_svb16_decode(hipStream_t s, const uint8_t *streams, const long long *stream_off,
                                        const long long *sample_off, int n_rows, int16_t *out, int *status) {
    hipLaunchKernelGGL(svb16_decode_kernel, dim3(n_rows), dim3(VBZ_THREADS), 0, s, streams, stream_off, sample_off,
                       out, status);
    return 0;
}
