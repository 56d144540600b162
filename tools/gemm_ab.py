#!/usr/bin/env python3
"""Round 4 A/B of the 256 x 256 tile GEMMs on RANDOM operands (debug library): gemm256_kernel (32x32x16 MFMA, dbg 0x1000)
against gemm256x_kernel (16x16x32 MFMA, dbg 0x2000 | column group << 4 | ablation bits; 0x400 / 0x800 = contiguous weight /
activation DMA sources), interleaved in one process.  MIBC_GX_STAGGER=<cycles> (debug library) sets the phase stagger.
    python tools/gemm_ab.py [rounds] [case-substring]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi
L = capi.dbg_lib()
L.mibc_debug_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
only = sys.argv[2] if len(sys.argv) > 2 else ""
cases = [
    ("sup43 head", 4096 * 1666, 4096, 1024, [("gx cg4", 0x2040), ("gx cg4 nt-stores", 0x2044), ("gx cg16 nt-stores", 0x2104), ("gx cg2 nt-stores", 0x2024)]),
    ("tx crf", 1 << 21, 4096, 512, [("g256 32x32x16", 0x1000), ("gx cg8", 0x2080), ("gx cg8 nt-stores", 0x2084), ("gx cg16 nt-stores", 0x2104)]),
    ("tx qkv-shaped (plain)", 1 << 20, 1536, 512, [("g256 32x32x16", 0x1000), ("gx", 0x2000), ("gx nt-stores", 0x2004)]),
]
for name, M, N, K, variants in cases:
    if only and only not in name:
        continue
    for r in range(rounds):
        for vn, dbg in variants:
            ms = C.c_float()
            rc = L.mibc_debug_gemm(M, N, K, dbg, 3, C.byref(ms))
            print(json.dumps({"case": name, "M": M, "N": N, "K": K, "variant": vn, "dbg": hex(dbg), "round": r, "rc": rc, "stagger": os.environ.get("MIBC_GX_STAGGER", "0"),
                              "ms": round(ms.value, 3), "tflops": round(2.0 * M * N * K / ms.value / 1e9, 1) if ms.value > 0 else None}), flush=True)
