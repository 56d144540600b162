"""Timing of the fused layer tail at the bench batch, with ablations (mode | dbg << 8)."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from dorado_amd import capi

import os
if os.environ.get("MIBC_LIB"):      # A/B of an alternative build (tools/txlayer_store16_ab.sh); a tool switch, not a product one
    capi.LIB_PATH = os.path.abspath(os.environ["MIBC_LIB"])
L = capi.dbg_lib() if not os.environ.get("MIBC_LIB") else capi.lib()
L.mibc_debug_txlayer_compare.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)] + \
    [C.POINTER(C.c_float)] * 5 + [C.c_void_p, C.c_void_p]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 1024
import numpy as np
L.mibc_device_count()
hip = C.CDLL("libamdhip64.so")
dbuf = C.c_void_p()
hip.hipMalloc(C.byref(dbuf), 128)
hip.hipMemset(dbuf, 0, 128)
L.mibc_debug_txlayer_trace.argtypes = [C.c_void_p]
for mode in [int(a, 0) for a in sys.argv[2:]] or [3, 3 | 0x4000, 1, 2, 2 | 0x200, 2 | 0x4200]:
    nd = C.c_longlong()
    f = [C.c_float() for _ in range(5)]
    rc = L.mibc_debug_txlayer_compare(R, 2048, mode, 3, C.byref(nd), *[C.byref(v) for v in f], None, None)
    fl = R * 2.0 * 512 * ((512 if mode & 1 else 0) + (3 * 2048 if mode & 2 else 0))
    L.mibc_debug_txlayer_trace(dbuf)
    g = [C.c_float() for _ in range(5)]
    L.mibc_debug_txlayer_compare(R, 2048, mode, 0, C.byref(nd), *[C.byref(v) for v in g], None, None)
    L.mibc_debug_txlayer_trace(None)
    st = np.zeros(16, np.uint64)
    hip.hipMemcpy(st.ctypes.data_as(C.c_void_p), dbuf, 128, 2)
    d = [int(st[i + 1]) - int(st[i]) if st[i + 1] and st[i] else None for i in range(5)]
    print("   cycle stamps (tile 1 of workgroup 0): out-proj", d[0], " norm 1", d[1], " fragment ring", d[2], " MLP", d[3] if d[3] else (int(st[4]) - int(st[0]) if st[4] else None), " norm 2", d[4])
    print("   slab 9: fc2 stage", int(st[9]) - int(st[8]), " fc1 (2 stages)", int(st[10]) - int(st[9]))
    print(f"mode {mode & 0xff} dbg {mode >> 8}: rc {rc} fused {f[3].value:.3f} ms ({fl / f[3].value / 1e9:.0f} TF)  five launches {f[4].value:.3f} ms  maxdiff {f[0].value:.4f}")
