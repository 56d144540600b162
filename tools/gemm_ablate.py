"""Debug build: gemm256 ablations on the K = 512 shapes (dbg = 0x1000 + bits: 1 no stores, 2 no epilogue, 4 A cache-resident, 8 no MFMA)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmibc_dbg.so")
L = capi.lib()
L.mibc_debug_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
for (M, N, K) in [(1 << 20, 4096, 512), (1 << 20, 512, 512), (1 << 20, 1536, 512)]:
    for bits in [0, 1, 2, 4, 6, 8, 10, 14]:
        ms = C.c_float()
        rc = L.mibc_debug_gemm(M, N, K, 0x1000 + bits if bits else 0, 5, C.byref(ms))
        print(f"M={M} N={N} K={K} ablate={bits:2d}: {ms.value:.3f} ms  {2.0*M*N*K/ms.value/1e9:.0f} TF (rc={rc})", flush=True)
