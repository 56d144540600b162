"""Debug helper: structure of the differences between the fused layer tail and the five-launch path."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from dorado_amd import capi

L = capi.dbg_lib()
L.mibc_debug_txlayer_compare.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)] + \
    [C.POINTER(C.c_float)] * 5 + [C.c_void_p, C.c_void_p]
import os
for mode in [int(a) for a in sys.argv[1:]] or [1]:
    R, FF = int(os.environ.get("TL_R", "512")), 2048
    a = np.zeros((R, 512), np.float16)
    b = np.zeros((R, 512), np.float16)
    nd = C.c_longlong()
    f = [C.c_float() for _ in range(5)]
    rc = L.mibc_debug_txlayer_compare(R, FF, mode, 1, C.byref(nd), *[C.byref(v) for v in f], a.ctypes.data, b.ctypes.data)
    print("mode", mode, "rc", rc, "ndiff", nd.value, "max", f[0].value, "rms", f[1].value, "amax", f[2].value)
    d = np.abs(a.astype(np.float32) - b.astype(np.float32))
    bad = d > 0.01
    print(" bad fraction", bad.mean(), "nan fused", np.isnan(a.astype(np.float32)).sum())
    print(" bad by row%128 block of 32 (wave):", [round(float(bad[(np.arange(R) % 128) // 32 == w].mean()), 3) for w in range(4)])
    print(" bad by row%8:", [round(float(bad[np.arange(R) % 8 == i].mean()), 3) for i in range(8)])
    tb = [(t, int(bad[128 * t:128 * t + 128].any(1).sum())) for t in range(R // 128) if bad[128 * t:128 * t + 128].any()]
    print(" tiles with bad rows (tile, bad rows):", tb[:40], "count", len(tb))
    if tb:
        t0 = tb[0][0]
        rows = np.nonzero(bad[128 * t0:128 * t0 + 128].any(1))[0]
        print(" bad rows in tile", t0, ":", rows.tolist())
        r0 = 128 * t0 + int(rows[0])
        print(" bad cols in that row:", np.nonzero(bad[r0])[0].tolist()[:64], "n", int(bad[r0].sum()))
        print(" fused", a[r0, :16].astype(np.float32), " ref", b[r0, :16].astype(np.float32))
    print(" bad by col//32:", [round(float(bad[:, 32 * c:32 * c + 32].mean()), 2) for c in range(16)])
    print(" bad by col%32:", [round(float(bad[:, np.arange(512) % 32 == c].mean()), 2) for c in range(32)])
    print(" fused row0[:16]", a[0, :16].astype(np.float32))
    print(" ref   row0[:16]", b[0, :16].astype(np.float32))
    # is the fused result a permutation of the reference within a row?
    print(" sorted-row match:", float(np.abs(np.sort(a.astype(np.float32), 1) - np.sort(b.astype(np.float32), 1)).max()))
    print(" corr row0:", float(np.corrcoef(a[0].astype(np.float32), b[0].astype(np.float32))[0, 1]))
