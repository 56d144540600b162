#!/usr/bin/env python3
"""SURVEY.md 8f-3 measurement: what variable chunk sizes buy on MI355X (hac@v4.3.0 shape).
A read set with a realistic length spread (log-normal, median 6 k samples, plus a tail of short reads) is cut
(a) into fixed 9996-sample chunks with repeat padding (BasecallerNode.cpp:432-440) and (b) with
generate_variable_chunks + row packing (2-step gaps); both are run as batches of N rows through the engine.
Reported: rows needed, useful samples per row-sample, time per batch in each mode (the masked LSTM instance and
the sample bitmap are the only extra work), useful Samples/s."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, hostapi, synth  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    cfg = config.hac_v43()
    ws = synth.make_weights(cfg, seed=1)
    N, T_in, st, ov = 4096, cfg.chunk_size, cfg.stride, cfg.overlap
    rng = np.random.default_rng(0)
    lens = np.concatenate([np.exp(rng.normal(np.log(6000), 0.9, 6000)), rng.uniform(300, 3000, 3000)]).astype(np.int64)
    lens = np.clip(lens, 200, 400000)
    useful = int(lens.sum())
    # (a) fixed chunks
    n_fixed = sum(len(hostapi.generate_chunks(int(L), T_in, st, ov)) for L in lens)
    # (b) variable chunks, first-fit in order
    table, row, fill = [], 0, 0
    for L in lens:
        for b, e in hostapi.generate_variable_chunks(int(L), T_in, st, ov):
            P = (e - b + st - 1) // st * st
            start = fill + 2 * st if fill else 0
            if start + P > T_in:
                row, fill, start = row + 1, 0, 0
            table.append((row, start, P))
            fill = start + P
    rows_var = row + 1
    eng = capi.Engine(cfg, ws)
    eng.reserve(N, T_in)
    T = eng.output_steps(T_in)
    L_ = capi.lib()
    d_in = eng.device_alloc(N * T_in * 2)
    d_out = eng.device_alloc(3 * N * T)
    x = synth.make_signal(64, T_in, seed=2)
    for r in range(0, N, 64):
        L_.mibc_memcpy_h2d(eng._h, C.c_void_p(d_in + r * T_in * 2), x.ctypes.data, x.nbytes)
    first = [c for c in table if c[0] < N]
    arr = eng._var_chunks(first)

    def timeit(fn, reps=3):
        fn()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        return (time.perf_counter() - t0) / reps

    t_fixed = timeit(lambda: L_.mibc_call_device(eng._h, d_in, N, T_in, C.byref(eng.opts), d_out))
    t_var = timeit(lambda: L_.mibc_call_device_var(eng._h, d_in, None, N, T_in, arr, len(first), C.byref(eng.opts), d_out))
    out = {"reads": int(lens.size), "useful_samples": useful, "rows_fixed": int(n_fixed), "rows_variable": int(rows_var),
           "fill_fixed": useful / (n_fixed * T_in), "fill_variable": useful / (rows_var * T_in),
           "ms_per_batch_fixed": t_fixed * 1e3, "ms_per_batch_variable": t_var * 1e3, "batch_rows": N,
           "chunks_in_variable_batch": len(first),
           "useful_samples_per_s_fixed": useful / (n_fixed / N * t_fixed),
           "useful_samples_per_s_variable": useful / (rows_var / N * t_var)}
    out["speedup"] = out["useful_samples_per_s_variable"] / out["useful_samples_per_s_fixed"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
