#!/usr/bin/env python3
"""Round 5 experiment: phase offset between the two waves of a SIMD (debug library switches, results unchanged):
MIBC_WS_STAGGER (wsgemm conv3 / head: waves 4-7 start every tile n x 1024 clocks late) and MIBC_X8_STAGGER (hac LSTM:
waves 4-7 start every time step n x 512 clocks late; runs on the MIBC_LSTM_DBG=8 copy of the kernel).  One child per setting."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pairs = [(0, 0), (2, 2), (4, 4), (6, 8), (8, 12), (3, 6), (0, 0)]
for ws, x8 in pairs:
    env = dict(os.environ, MIBC_WS_STAGGER=str(ws), MIBC_X8_STAGGER=str(x8), MIBC_LSTM_DBG="8")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_times.py"), "--lib", "dbg", "--model", "hac", "--steps", "3"],
                       env=env, capture_output=True, text=True, timeout=400)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"ws_stagger {ws} x8_stagger {x8}: conv {d['conv']} lstm {d['lstm']} {d['lstm_layer']} head {d['head']} decode {d['decode']} total {d['total']}", flush=True)
    except Exception:
        print(f"ws_stagger {ws} x8_stagger {x8}: FAILED {r.stderr[-300:]}", flush=True)
