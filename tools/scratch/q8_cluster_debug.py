#!/usr/bin/env python3
"""Debug aid (GPU): one int8 cluster LSTM layer against an integer emulation in numpy.  tiny(512) with 2 layers: layer 0 f16
(cluster kernel), layer 1 int8 -> f16 (Q8 == 2).  Prints rms differences of the LSTM stack output (tap 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dorado_amd import capi, config, synth

def sig(x): return 1.0 / (1.0 + np.exp(-x))

def layer(xin, Wih, Whh, b, rev, quant):
    T, N, C = xin.shape
    W = np.concatenate([Wih, Whh], 1).astype(np.float32)
    if quant:
        q, scale = capi.quantize_lstm_weights(Wih, Whh)
        Wq = q.astype(np.float32)
        deq = (1.0 / (127.0 * scale)).astype(np.float32)
        xq = np.clip(np.rint(np.clip(xin, -1, 1) * 127.0), -127, 127).astype(np.float32)
    out = np.zeros((T, N, C), np.float32); h = np.zeros((N, C), np.float32); c = np.zeros((N, C), np.float32)
    for s in range(T):
        t = T - 1 - s if rev else s
        if quant:
            hq = np.rint(h * 127.0).astype(np.float32)
            pre = (np.concatenate([xq[t], hq], 1) @ Wq.T) * deq[None, :] + b[None, :]
        else:
            pre = np.concatenate([xin[t], h], 1) @ W.T + b[None, :]
        i, f, g, o = pre[:, :C], pre[:, C:2 * C], pre[:, 2 * C:3 * C], pre[:, 3 * C:]
        c = sig(f) * c + sig(i) * np.tanh(g); h = sig(o) * np.tanh(c); out[t] = h
    return out

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = config.tiny(C_, 5); cfg.lstm_layers = L_
ws = synth.make_weights(cfg, seed=21)
MODE = sys.argv[3] if len(sys.argv) > 3 else ""
for l in range(L_):
    b0 = 2 * len(cfg.convs) + 4 * l
    if "nobias" in MODE:
        ws[b0 + 2] = 0 * ws[b0 + 2]; ws[b0 + 3] = 0 * ws[b0 + 3]
    if "nohh" in MODE:
        ws[b0 + 1] = 0 * ws[b0 + 1]
    if "noih" in MODE and l >= 1:
        ws[b0] = 0 * ws[b0]
    if "ih1" in MODE:
        ws[b0] = ws[b0] / 8.0
print("mode", MODE)
x = synth.make_signal(256, 306, seed=22)
base = 2 * len(cfg.convs)
cfg1 = config.tiny(C_, 5); cfg1.lstm_layers = 1
ws1 = ws[:base + 4] + ws[base + 4 * L_:]           # convs + first LSTM layer + head: the conv output survives in xa (tap 2)
e = capi.Engine(cfg1, ws1, taps=True); e.forward(x); T = e.output_steps(306)
conv = e.tap(2, (T, 256, C_), np.float16).astype(np.float32); e.close()
e = capi.Engine(cfg, ws, taps=True); e.forward(x)
h16 = e.tap(3, (T, 256, C_), np.float16).astype(np.float32); e.close()
cfg.lstm_quant = True
e = capi.Engine(cfg, ws, taps=True); e.forward(x); h8 = e.tap(3, (T, 256, C_), np.float16).astype(np.float32); e.close()
cur16 = conv; cur8 = conv
for l in range(L_):
    Wih, Whh, b = ws[base + 4 * l].astype(np.float16).astype(np.float32), ws[base + 4 * l + 1].astype(np.float16).astype(np.float32), ws[base + 4 * l + 2] + ws[base + 4 * l + 3]
    rev = (l % 2 == 0)
    cur16 = layer(cur16[:, :8], Wih, Whh, b, rev, False) if l else layer(conv[:, :8], Wih, Whh, b, rev, False)
    if l == 0:
        cur8 = cur16
    else:
        cur8 = layer(cur8, ws[base + 4 * l], ws[base + 4 * l + 1], b, rev, True)
r = lambda a, b_: float(np.sqrt(((a - b_) ** 2).mean()))
print(f"C={C_} layers={L_}: device f16 vs numpy f32: {r(h16[:, :8], cur16):.5f}; device int8 vs numpy int8 emulation: {r(h8[:, :8], cur8):.5f}; "
      f"numpy int8 vs numpy f32: {r(cur8, cur16):.5f}; device int8 vs device f16: {r(h8, h16):.5f}")
for t in (0, 1, T // 2, T - 1):
    print("  t", t, "rms dev8-emu8", f"{r(h8[t, :8], cur8[t]):.5f}", " first units dev", np.round(h8[t, 0, :6], 3), "emu", np.round(cur8[t, 0, :6], 3))
for u0 in range(0, C_, 64):
    print("  units", u0, "rms dev8-emu8", f"{r(h8[:, :8, u0:u0 + 64], cur8[:, :, u0:u0 + 64]):.5f}")
