#!/usr/bin/env python3
"""SURVEY.md 8f-1 measurement: the device side of ScalerNode on MI355X.
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
