#!/usr/bin/env python3
"""Round 5 experiment (debug library switches; cache policy only, results unchanged): non-temporal loads / stores for data that
is touched once.  MIBC_WS_DBG=8: wsgemm (conv3 / head) activation tile loads non-temporal; MIBC_Q8_NT=1: int8 LSTM kernel, x_t loads
and h_t stores non-temporal (as adopted for lstm_layer_x8_kernel, profiles/r05_h_x8_nt_ab.log).  One child per setting, alternating;
stage times in ms (HIP events)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(model, batch, quant, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_times.py"), "--lib", "dbg", "--model", model, "--batch", str(batch),
                        "--steps", "3", "--quant", str(quant)], env=env, capture_output=True, text=True, timeout=400)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{model} quant {quant} {env_extra}: conv {d['conv']} lstm {d['lstm']} {d['lstm_layer']} head {d['head']} decode {d['decode']} total {d['total']}", flush=True)
    except Exception:
        print(f"{model} quant {quant} {env_extra}: FAILED {r.stderr[-300:]}", flush=True)


for _ in range(4):
    run("hac", 16384, 0, {"MIBC_WS_DBG": "0"})
    run("hac", 16384, 0, {"MIBC_WS_DBG": "8"})
for _ in range(4):
    run("hac", 16384, 1, {"MIBC_Q8_NT": "0"})
    run("hac", 16384, 1, {"MIBC_Q8_NT": "1"})
for _ in range(2):
    run("sup", 8192, 0, {"MIBC_WS_DBG": "0"})
    run("sup", 8192, 0, {"MIBC_WS_DBG": "8"})
