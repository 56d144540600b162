import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, synth
cfg = config.tiny(384, 3); cfg.lstm_layers = 1
ws = synth.make_weights(cfg, seed=384)
N, T_in = 2048, 36
x = synth.make_signal(N, T_in, seed=385)
eng = capi.Engine(cfg, ws)
def out(xx):
    eng.forward(xx); T = eng.output_steps(xx.shape[1])
    return eng.tap(3, (T, xx.shape[0], 384), np.float16)
a_x8 = np.concatenate([out(x[i:i+1024]) for i in range(0, N, 1024)], axis=1)
eng.set_ws_min_rows(2048)
a_ws = out(x)
T = a_ws.shape[0]
bad = (a_ws.view(np.uint16) != a_x8.view(np.uint16))
print("T", T, "total bad", bad.sum(), "of", bad.size)
for t in range(T - 1, -1, -1):   # layer 0 runs in reverse: step 0 = t = T-1
    b = bad[t]
    if b.sum() == 0:
        print("t", t, "clean"); continue
    rows = np.nonzero(b.any(axis=1))[0]; cols = np.nonzero(b.any(axis=0))[0]
    print("t", t, "bad", b.sum(), "rows", len(rows), rows[:12], "cols", len(cols), cols[:16], "row%16 hist", np.bincount(rows % 16, minlength=16).tolist(), "col//16 hist", np.bincount(cols // 16, minlength=24).tolist())
    d = np.abs(a_ws[t].astype(np.float32) - a_x8[t].astype(np.float32))
    print("   maxdiff", d.max(), "example", a_ws[t][b][:4], a_x8[t][b][:4])
    break
