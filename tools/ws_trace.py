"""Debug build only: phase timeline (s_memtime stamps) of the weight-stationary LSTM kernel, MIBC_WS_LSTM_DBG=128|192.
    MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=128 python tools/ws_trace.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, synth  # noqa: E402

capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmibc_dbg.so")
cfg = config.hac_v43()
cfg.lstm_layers = 1
eng = capi.Engine(cfg, synth.make_weights(cfg, seed=42))
n, t_in = 16384, 2400
x = np.tile(synth.make_signal(128, t_in, seed=1), (n // 128, 1))
eng.forward(x)
buf = (C.c_ulonglong * 256)()
assert capi.lib().mibc_debug_ws_trace(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(16, 2, 8)
names_x = ["bar", "stores", "dma", "mfma", "handw", "top", "vmw"]
for it in range(16):
    xs, hs = a[it, 0], a[it, 1]
    base = xs[5]
    print("it %2d  x: top->vmcnt %5d ->barrier %5d | stores %5d | dma %5d | mfma %5d | handoff %5d ||  h: top %6d bar %6d mfma %5d gates+write %5d"
          % (it, xs[6] - xs[5], xs[0] - xs[6], xs[1] - xs[0], xs[2] - xs[1], xs[3] - xs[2], xs[4] - xs[3],
             hs[5] - base, hs[0] - base, hs[2] - hs[0], hs[4] - hs[2]))
    if it < 15:
        print("        iteration length (x top to next top): %d" % (a[it + 1, 0, 5] - xs[5]))
