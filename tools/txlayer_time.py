"""Timing of the fused layer tail at the bench batch, with ablations (mode | dbg << 8)."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from dorado_amd import capi

L = capi.lib()
L.mibc_debug_txlayer_compare.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)] + \
    [C.POINTER(C.c_float)] * 5 + [C.c_void_p, C.c_void_p]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 1024
for mode in [int(a, 0) for a in sys.argv[2:]] or [3, 3 | 0x100, 3 | 0x200, 1, 2]:
    nd = C.c_longlong()
    f = [C.c_float() for _ in range(5)]
    rc = L.mibc_debug_txlayer_compare(R, 2048, mode, 3, C.byref(nd), *[C.byref(v) for v in f], None, None)
    fl = R * 2.0 * 512 * ((512 if mode & 1 else 0) + (3 * 2048 if mode & 2 else 0))
    print(f"mode {mode & 0xff} dbg {mode >> 8}: rc {rc} fused {f[3].value:.3f} ms ({fl / f[3].value / 1e9:.0f} TF)  five launches {f[4].value:.3f} ms  maxdiff {f[0].value:.4f}")
