#!/usr/bin/env python3
"""Experiment: R independent engines (own stream + workspace, N/R chunks each) on ONE GPU, the way the reference
runs num_runners = 2 per device: while one runner's batch is in its HBM/VALU-bound tail (head, decoder) or front
(convolutions), the others' LSTM kernels keep the matrix pipes busy.  Prints samples/s per R."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, synth  # noqa: E402


def run(R, N_total, steps, stagger):
    cfg = config.hac_v43()
    ws = synth.make_weights(cfg, seed=42)
    t_in = cfg.chunk_size
    n = N_total // R
    engs, bufs = [], []
    base = synth.make_signal(256, t_in, seed=1)
    x = np.tile(base, ((n + 255) // 256, 1))[:n]
    for r in range(R):
        e = capi.Engine(cfg, ws)
        e.reserve(n, t_in)
        T = e.output_steps(t_in)
        d_in, d_out = e.device_alloc(x.nbytes), e.device_alloc(3 * n * T)
        e.h2d(d_in, x)
        engs.append(e)
        bufs.append((d_in, d_out))
    for e, (a, b) in zip(engs, bufs):          # warm-up
        e.call_device(a, n, t_in, b)
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        for r, (e, (a, b)) in enumerate(zip(engs, bufs)):
            if k == 0 and r > 0 and stagger > 0:
                time.sleep(stagger / R)
            e.call_device(a, n, t_in, b)
    for e in engs:
        e.sync()
    el = time.perf_counter() - t0
    for e, (a, b) in zip(engs, bufs):
        e.device_free(a)
        e.device_free(b)
        e.close()
    return N_total * t_in * steps / el, el / steps * 1e3


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    out = {}
    for R, stag in [(1, 0.0), (2, 0.0), (2, 0.35), (4, 0.35)]:
        v, ms = run(R, 16384, steps, stag)
        out[f"R{R}_stagger{stag}"] = {"samples_per_s": v, "ms_per_global_step": ms}
        print(R, stag, f"{v:.4g}", f"{ms:.1f} ms", flush=True)
    print(json.dumps(out))
