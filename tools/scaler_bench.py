#!/usr/bin/env python3
"""SURVEY.md 8f-1 / 8f-2 measurement: the device side of ScalerNode and of the POD5 signal decode on MI355X.
  * mibc_svb16_decode     StreamVByte-16 + zig-zag + delta stage of POD5 VBZ rows
  * mibc_scale_reads      HBM-bound, 2 B read + 2 B written per sample
  * mibc_scaler_stats     one pass over the samples (2 B/sample) + LDS histogram atomics
  * mibc_call_device_i16  the whole hot path fed with raw int16 chunks vs pre-scaled f16 chunks
                          (the fused map must cost nothing: conv1 reads 2 B/sample either way)
Prints one JSON object; timings are HIP-event free wall times around synchronised calls."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, synth  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    cfg = config.hac_v43()
    ws = synth.make_weights(cfg, seed=1)
    eng = capi.Engine(cfg, ws)
    L = capi.lib()
    out = {}
    # ---- whole reads: 4096 reads x 250k samples = 1.02e9 samples
    n_reads, rl = 4096, 250_000
    rng = np.random.default_rng(0)
    one = (480 + 90 * rng.standard_normal(rl)).astype(np.int16)
    total = n_reads * rl
    d_sig = eng.device_alloc(total * 2)
    d_out = eng.device_alloc(total * 2)
    for r in range(0, n_reads, 256):  # fill by tiles of the same read (content does not matter for timing)
        blk = np.tile(one, min(256, n_reads - r))
        L.mibc_memcpy_h2d(eng._h, C.c_void_p(d_sig + r * rl * 2), blk.ctypes.data, blk.nbytes)
    off = (np.arange(n_reads + 1, dtype=np.int64) * rl)
    d_off = eng.device_alloc(off.nbytes)
    eng.h2d(d_off, off)
    d_ss = eng.device_alloc(n_reads * 8)
    d_raw = eng.device_alloc(n_reads * 8)
    p = (C.c_float * 4)(0.2, 0.9, 0.51, 0.53)

    def timeit(fn, reps=5):
        fn()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        return (time.perf_counter() - t0) / reps

    t = timeit(lambda: L.mibc_scaler_stats(eng._h, d_sig, d_off, n_reads, 0, p, d_ss, d_raw))
    out["scaler_stats_quantile"] = {"samples_per_s": total / t, "ms": t * 1e3, "GB_per_s": total * 2 / t / 1e9}
    t = timeit(lambda: L.mibc_scaler_stats(eng._h, d_sig, d_off, n_reads, 1, None, d_ss, d_raw))
    out["scaler_stats_med_mad"] = {"samples_per_s": total / t, "ms": t * 1e3, "GB_per_s": total * 2 / t / 1e9}
    t = timeit(lambda: L.mibc_scale_reads(eng._h, d_sig, d_off, n_reads, d_ss, d_out))
    out["scale_reads"] = {"samples_per_s": total / t, "ms": t * 1e3, "GB_per_s": total * 4 / t / 1e9,
                          "algorithmic_bytes_per_sample": 4, "frac_of_8TBs": total * 4 / t / 8e12}
    for q in (d_sig, d_out, d_off, d_ss, d_raw):
        eng.device_free(q)
    # ---- POD5 VBZ svb16 stage: 4096 rows x 102400 samples (the row size MinKNOW writes), ~12 % two-byte values
    rows, rs = 4096, 102400
    d = rng.integers(-60, 60, rs).astype(np.int64)
    d[rng.random(rs) < 0.12] *= 9                       # some deltas need two bytes
    xw = np.cumsum(d).astype(np.int16)
    dz = np.diff(np.concatenate([[0], xw.astype(np.int64)])).astype(np.int16).view(np.uint16)
    z = ((dz << 1) ^ (0 - (dz >> 15))).astype(np.uint16)
    two = z >= 256
    keys = np.packbits(two, bitorder="little")
    lens = 1 + two.astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    data = np.zeros(int(lens.sum()), np.uint8)
    data[offs] = (z & 0xff).astype(np.uint8)
    data[offs[two] + 1] = (z[two] >> 8).astype(np.uint8)
    stream = np.concatenate([keys, data])
    sl = stream.size
    d_st = eng.device_alloc(rows * sl)
    for r in range(rows):
        L.mibc_memcpy_h2d(eng._h, C.c_void_p(d_st + r * sl), stream.ctypes.data, sl) if r < 64 else None
    for r in range(64, rows, 64):   # replicate the first 64 rows on the device side through the host copy
        blk = np.tile(stream, 64)
        L.mibc_memcpy_h2d(eng._h, C.c_void_p(d_st + r * sl), blk.ctypes.data, blk.nbytes)
    so = (np.arange(rows + 1, dtype=np.int64) * sl)
    no = (np.arange(rows + 1, dtype=np.int64) * rs)
    d_so, d_no = eng.device_alloc(so.nbytes), eng.device_alloc(no.nbytes)
    eng.h2d(d_so, so)
    eng.h2d(d_no, no)
    d_x = eng.device_alloc(rows * rs * 2)
    d_stat = eng.device_alloc(rows * 4)
    t = timeit(lambda: L.mibc_svb16_decode(eng._h, d_st, d_so, d_no, rows, d_x, d_stat))
    chk = np.zeros(rs, np.int16)
    eng.d2h(chk, d_x + (rows - 1) * rs * 2)
    stat = np.zeros(rows, np.int32)
    eng.d2h(stat, d_stat)
    out["svb16_decode"] = {"samples_per_s": rows * rs / t, "ms": t * 1e3,
                           "stream_bytes_per_sample": sl / rs, "GB_per_s": rows * (sl + 2 * rs) / t / 1e9,
                           "frac_of_8TBs": rows * (sl + 2 * rs) / t / 8e12,
                           "verified": bool((chk == xw).all() and not stat.any())}
    for q in (d_st, d_so, d_no, d_x, d_stat):
        eng.device_free(q)
    # ---- fused path: hac batch, raw int16 in vs scaled f16 in
    N, T_in = 16384, cfg.chunk_size
    eng.reserve(N, T_in)
    T = eng.output_steps(T_in)
    d_in = eng.device_alloc(N * T_in * 2)
    d_o = eng.device_alloc(3 * N * T)
    x = (480 + 95 * synth.make_signal(64, T_in, seed=2).astype(np.float32)).astype(np.int16)
    for r in range(0, N, 64):
        L.mibc_memcpy_h2d(eng._h, C.c_void_p(d_in + r * T_in * 2), x.ctypes.data, x.nbytes)
    ss = np.tile(np.array([[480.0, 95.0]], np.float32), (N, 1))
    d_ss2 = eng.device_alloc(ss.nbytes)
    eng.h2d(d_ss2, ss)
    t_i16 = timeit(lambda: L.mibc_call_device_i16(eng._h, d_in, d_ss2, N, T_in, C.byref(eng.opts), d_o), reps=3)
    t_f16 = timeit(lambda: L.mibc_call_device(eng._h, d_in, N, T_in, C.byref(eng.opts), d_o), reps=3)
    out["hot_path_hac_n16384"] = {"ms_int16_in_fused_scaling": t_i16 * 1e3, "ms_f16_in": t_f16 * 1e3,
                                  "samples_per_s_int16_in": N * T_in / t_i16}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
