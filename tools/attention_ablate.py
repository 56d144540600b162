#!/usr/bin/env python3
"""Timing ablations of window_attention_v3_kernel (debug library): N chunks x 1024 tokens x 8 heads, random q / k / v.
dbg bits: 1 no exponentials, 2 no PV product, 4 no QK^T product, 8 ring not refilled (no loads / LDS writes / ring barriers)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi
L = capi.dbg_lib()
L.mibc_debug_attention_compare.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_longlong)] + [C.POINTER(C.c_float)] * 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for rnd in range(2):
    for dbg in (0, 64, 8, 32, 16, 63):
        nd, t3, t2 = C.c_longlong(), C.c_float(), C.c_float()
        rc = L.mibc_debug_attention_compare(N, 1024, 8, 127, 128, 5 | (dbg << 16), C.byref(nd), C.byref(t3), C.byref(t2), None, None)
        print(f"round {rnd} dbg {dbg:2d}: ring kernel {t3.value * 1024 / N:.3f} ms per 1024 chunks (rc {rc}); re-staging kernel {t2.value * 1024 / N:.3f}")
