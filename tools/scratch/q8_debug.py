"""Debug helper: int8 LSTM path vs f16 path, layer by layer (tap 3 = LSTM stack output)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from dorado_amd import capi, config, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for layers in (2, 3):
    cfg = config.tiny(C, 4)
    cfg.lstm_layers = layers
    ws = synth.make_weights(cfg, seed=11)
    x = synth.make_signal(64, 606, seed=12)
    outs = {}
    for q in (False, True):
        cfg.lstm_quant = q
        e = capi.Engine(cfg, ws, taps=True)
        sc = e.forward(x)
        T = e.output_steps(606)
        outs[q] = e.tap(3, (T, 64, C), np.float16).astype(np.float32)
        e.close()
    d = np.abs(outs[True] - outs[False])
    print("layers", layers, "lstm out: max", d.max(), "rms", np.sqrt((d ** 2).mean()), "ref amax", np.abs(outs[False]).max(),
          "q amax", np.abs(outs[True]).max())
    print("  by t (first/last 3):", d.mean(axis=(1, 2))[:3], d.mean(axis=(1, 2))[-3:])
    print("  by hidden%16:", np.round(d.mean(axis=(0, 1)).reshape(-1, 16).mean(0), 3))
    print("  by row%16:", np.round(d.mean(axis=(0, 2)).reshape(-1, 16).mean(0), 3))
    print("  sample q  :", outs[True][5, 3, :8])
    print("  sample f16:", outs[False][5, 3, :8])
