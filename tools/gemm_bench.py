#!/usr/bin/env python3
"""GEMM microbenchmark through libmibc's debug entry: prints TFLOP/s for the shapes of the hot path."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi
L = capi.dbg_lib()
L.mibc_debug_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
shapes = [(1 << 20, 1536, 512), (1 << 20, 512, 512), (1 << 20, 4096, 512), (1 << 20, 512, 2048), (1 << 21, 4096, 512), (1 << 22, 1024, 384),
          (4096 * 1666, 4096, 1024)]   # dbg 0 = production path (gemm256 where it applies), 256 = gemm_dma_kernel
for dbg in [int(x) for x in (sys.argv[1:] or ["0"])]:
    for (M, N, K) in shapes:
        ms = C.c_float()
        rc = L.mibc_debug_gemm(M, N, K, dbg, 5, C.byref(ms))
        print(f"dbg={dbg} M={M} N={N} K={K}: {ms.value:.3f} ms  {2.0*M*N*K/ms.value/1e9:.0f} TFLOP/s  (rc={rc})")
