"""gemm256_kernel (persistent 256 x 256 tile GEMM, gemm256.hip) == gemm_dma_kernel bit for bit on the shapes of
the hot path: sup head (K = 1024), transformer out-proj / FC2 / upsample / CRF (plain, bias), QKV (rotary epilogue +
transposed V), FC1 (SwiGLU).  Same MFMA shape, same k order, same epilogue arithmetic -> identical halfs; the
BASELINE-size parity tests then pin the whole path to the reference."""
import ctypes as C

import pytest

from dorado_amd import capi

pytestmark = pytest.mark.gpu

SHAPES = [
    # M, N, K, epi, act, bias, rope_T
    (8192, 4096, 1024, 0, -1, 0, 0),      # sup@v4.3 head
    (8192, 1024, 1024, 0, 3, 1, 0),       # 5*tanh head with bias
    (4096, 512, 512, 0, -1, 1, 0),        # out-proj (+bias)
    (4096, 512, 2048, 0, -1, 0, 0),       # FC2
    (4096, 4096, 512, 0, -1, 0, 0),       # CRF
    (4096, 1536, 512, 1, -1, 0, 1024),    # QKV + rotary + V^T
    (4096, 4096, 512, 2, -1, 0, 0),       # FC1 + SwiGLU
    (2048 + 300, 512, 512, 0, -1, 0, 0),  # ragged last row tile
]


@pytest.mark.parametrize("M,N,K,epi,act,bias,rope_T", SHAPES)
def test_gemm256_bit_identical_to_gemm_dma(M, N, K, epi, act, bias, rope_T):
    L = capi.dbg_lib()
    L.mibc_debug_gemm_compare.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_longlong), C.POINTER(C.c_float),
                                                            C.POINTER(C.c_float), C.POINTER(C.c_float)]
    nd, md, t256, t128 = C.c_longlong(), C.c_float(), C.c_float(), C.c_float()
    rc = L.mibc_debug_gemm_compare(M, N, K, epi, act, bias, rope_T, 2, C.byref(nd), C.byref(md), C.byref(t256), C.byref(t128))
    assert rc == 0
    print(f"M={M} N={N} K={K} epi={epi}: differing halfs {nd.value} (max {md.value:.5f}); "
          f"{t256.value:.3f} ms vs {t128.value:.3f} ms")
    assert nd.value == 0
