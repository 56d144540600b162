"""Debug helper: distribution and location of the differences fused vs five launches (mode 3)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from dorado_amd import capi

L = capi.dbg_lib()
L.mibc_debug_txlayer_compare.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)] + \
    [C.POINTER(C.c_float)] * 5 + [C.c_void_p, C.c_void_p]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R, FF = int(sys.argv[2]) if len(sys.argv) > 2 else 38400, 2048
a = np.zeros((R, 512), np.float16)
b = np.zeros((R, 512), np.float16)
nd = C.c_longlong()
f = [C.c_float() for _ in range(5)]
rc = L.mibc_debug_txlayer_compare(R, FF, mode, 1, C.byref(nd), *[C.byref(v) for v in f], a.ctypes.data, b.ctypes.data)
print("mode", mode, "rc", rc, "ndiff", nd.value, "max", f[0].value, "rms", f[1].value)
af, bf = a.astype(np.float32), b.astype(np.float32)
d = np.abs(af - bf)
ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(bf), 2.0 ** -14))) - 10), 2.0 ** -24)
du = d / ulp
for k in (0.5, 1.5, 2.5, 4.5, 8.5, 16.5, 32.5):
    print(f" > {k} ulp: {(du > k).sum()}")
rows_bad = np.nonzero((du > 4.5).any(1))[0]
print(" rows with >4.5 ulp:", len(rows_bad), rows_bad[:40])
if len(rows_bad):
    r = rows_bad[0]
    print(" row", r, "n elements differing", (d[r] > 0).sum(), "max ulp", du[r].max(), "mean rel diff", float((d[r] / (np.abs(bf[r]) + 1e-3)).mean()))
    rel = (af[r] - bf[r]) / np.where(np.abs(bf[r]) > 0.05, bf[r], np.nan)
    print(" signed relative diff of the row (percentiles):", np.nanpercentile(rel, [1, 25, 50, 75, 99]))
    print(" row rms fused", np.sqrt((af[r] ** 2).mean()), "ref", np.sqrt((bf[r] ** 2).mean()))
    rr = np.array([np.sqrt((af[q] ** 2).mean()) / np.sqrt((bf[q] ** 2).mean()) for q in rows_bad[:200]])
    print(" rms ratio fused/ref of bad rows: min", rr.min(), "max", rr.max())
    print(" bad rows %128:", np.bincount(rows_bad % 128, minlength=128).nonzero()[0][:64])
    print(" bad rows //128 (tiles) distinct:", len(set(rows_bad // 128)), "of", R // 128)
