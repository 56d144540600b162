"""Scratch: where do the fused conv3 int8 epilogue and the separate conversion pass differ?  (debug library, MIBC_FUSE_Q8,
MIBC_STOP_AFTER_CONV3: tap 3 = the int8 rows the LSTM stack would read)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from dorado_amd import capi, config, synth
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmibc_dbg.so")
cfg = config.tiny(384, 4)
cfg.lstm_quant = sys.argv[3] == "1"
eng = capi.Engine(cfg, synth.make_weights(cfg, seed=41))
x = synth.make_signal(128, 1206, seed=42)
eng.forward(x)
T = eng.output_steps(1206)
if cfg.lstm_quant:
    np.save(sys.argv[2], eng.tap(3, (T, 128, 384), np.int8))
else:
    np.save(sys.argv[2], eng.tap(3, (T, 128, 384), np.float16))
"""
out = []
for fuse, q in (("0", "1"), ("1", "1"), ("0", "0")):
    p = f"/tmp/c3_{len(out)}.npy"
    r = subprocess.run([sys.executable, "-c", S, ROOT, p, q], env=dict(os.environ, MIBC_FUSE_Q8=fuse, MIBC_STOP_AFTER_CONV3="1"),
                       capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr[-300:])
    out.append(np.load(p))
a, b, f = out
d = np.argwhere(a != b)
print("separate vs fused int8 conv3 rows: differing elements", len(d), "of", a.size)
ff = f.astype(np.float32)
want = np.rint(np.clip(ff, -1, 1) * 127).astype(np.int8)
print("separate vs numpy(round(127 f16)):", int((a != want).sum()), " fused vs numpy:", int((b != want).sum()))
for t, n, c in d[:20]:
    print((t, n, c), "separate", a[t, n, c], "fused", b[t, n, c], "f16", float(f[t, n, c]), "x127", float(ff[t, n, c]) * 127)
