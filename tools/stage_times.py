#!/usr/bin/env python3
"""Print hipEvent stage times of one hot-path call (hac shape by default).
    python tools/stage_times.py [--batch N] [--steps K] [--model hac]
Env knobs are read by libmibc (e.g. MIBC_LSTM_ABLATE, MIBC_DECODE_SUB)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--model", default="hac")
ap.add_argument("--tin", type=int, default=0)
ap.add_argument("--profile-level", type=int, default=1, help="2 = also roctx ranges around the stages (rocprofv3 --marker-trace)")
ap.add_argument("--quant", type=int, default=0, help="1 = lstm_quant (the int8 LSTM path)")
ap.add_argument("--lib", default="", help="'dbg' = dorado_amd/libmibc_dbg.so (make -C dorado_amd/csrc debug): MIBC_* switches")
a = ap.parse_args()
if a.lib == "dbg":
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmibc_dbg.so")
elif a.lib:
    capi.LIB_PATH = os.path.abspath(a.lib)        # an A/B build of the library (tool switch, not a product one)
cfg = {"hac": config.hac_v43, "sup": config.sup_v43, "sup5": config.sup_v50, "fast": config.fast_v43}.get(a.model, lambda: config.tiny(128, 4))()
if a.quant:
    cfg.lstm_quant = True
t_in = a.tin or cfg.chunk_size
eng = capi.Engine(cfg, synth.make_weights(cfg, seed=42))
T = eng.output_steps(t_in)
n = a.batch
eng.reserve(n, t_in)
base = synth.make_signal(min(n, 128), t_in, seed=1)
x = np.tile(base, ((n + base.shape[0] - 1) // base.shape[0], 1))[:n]
d_in = eng.device_alloc(x.nbytes)
d_out = eng.device_alloc(3 * n * T)
eng.h2d(d_in, x)
eng.set_profile(a.profile_level)
res = None
for _ in range(a.steps):
    eng.call_device(d_in, n, t_in, d_out)
    res = eng.stage_ms()
res["lstm_layer"] = [round(v, 2) for v in res["lstm_layer"][: max(1, cfg.lstm_layers)]]
res["samples_per_s"] = round(n * t_in / (res["total"] * 1e-3))
res = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("MIBC_")}
print(json.dumps(res))
