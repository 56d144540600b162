#!/usr/bin/env python3
"""Round 5 experiment (debug library switches; cache policy only, results unchanged): non-temporal loads / stores for data that
is touched once.  MIBC_DEC_NT bits: 1 = k1 bwd_scan2 (score loads, guide stores), 2 = k2 beam search (score + guide loads),
4 = k3 posts_qual (score + guide loads).  MIBC_LSTM_DBG: 8 = plain copy of the hac LSTM kernel, 24 = nt x_t loads,
40 = nt h_t stores, 56 = both.  One child per setting; stage times in ms (HIP events)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = sys.argv[1] if len(sys.argv) > 1 else "hac"
batch = {"hac": "16384", "sup": "8192", "sup5": "1024"}[model]
runs = [("0", "8"), ("0", "56")] * 5 if model == "hac" else [("0", "0")]
for nt, x8 in runs:
    env = dict(os.environ, MIBC_DEC_NT=nt)
    if x8 != "0":
        env["MIBC_LSTM_DBG"] = x8
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_times.py"), "--lib", "dbg", "--model", model, "--batch", batch, "--steps", "3"],
                       env=env, capture_output=True, text=True, timeout=400)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{model} dec_nt {nt} lstm_dbg {x8}: conv {d['conv']} lstm {d['lstm']} {d['lstm_layer']} head {d['head']} decode {d['decode']} total {d['total']}", flush=True)
    except Exception:
        print(f"{model} dec_nt {nt} lstm_dbg {x8}: FAILED {r.stderr[-300:]}", flush=True)
