"""The persistent 256 x 256 tile GEMMs of the hot path (pytest -m gpu), on its shapes: sup head (K = 1024), transformer
out-proj / upsample / CRF (plain, bias), QKV (rotary epilogue + transposed V), and — debug-library paths of the unfused
transformer layer — FC2 (K = 2048) and FC1 (SwiGLU).

  gemm256_kernel  (gemm256.hip, v_mfma_f32_32x32x16_f16): same MFMA shape, k order and epilogue arithmetic as
                  gemm_dma_kernel -> bit-identical halfs (asserted).
  gemm256x_kernel (gemm256x.hip, round 4, v_mfma_f32_16x16x32_f16: the production choice for the plain / rotary epilogues
                  with K <= 1024): a different summation tree inside the instruction, so the contract is numeric — against
                  gemm_dma_kernel max-abs <= 0.008 (outputs reach +-5: one f16 ulp there is 0.0039) and against an f64 host
                  product on 4096 sampled outputs <= 0.005 (half an ulp of the f16 result + accumulation).
The BASELINE-size parity tests then pin the whole path to the reference."""
import ctypes as C

import pytest

from dorado_amd import capi

pytestmark = pytest.mark.gpu

SHAPES = [
    # M, N, K, epi, act, bias, rope_T
    (8192, 4096, 1024, 0, -1, 0, 0),      # sup@v4.3 head (column groups: 4 x 4 tiles)
    (8192, 1024, 1024, 0, 3, 1, 0),       # 5*tanh head with bias
    (4096, 512, 512, 0, -1, 1, 0),        # out-proj (+bias)
    (4096, 512, 2048, 0, -1, 0, 0),       # FC2
    (4096, 4096, 512, 0, -1, 0, 0),       # CRF (column groups: 2 x 8 tiles)
    (4096, 1536, 512, 1, -1, 0, 1024),    # QKV + rotary + V^T
    (4096, 4096, 512, 2, -1, 0, 0),       # FC1 + SwiGLU
    (2048 + 300, 512, 512, 0, -1, 0, 0),  # ragged last row tile
    (256 * 37 + 11, 4096, 1024, 0, -1, 0, 0),   # column groups with a ragged, odd number of row tiles
]


def _run(M, N, K, epi, act, bias, rope_T, dbg0):
    L = capi.dbg_lib()
    L.mibc_debug_gemm_compare.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_longlong)] + [C.POINTER(C.c_float)] * 4
    nd, md, t256, t128, ef = C.c_longlong(), C.c_float(), C.c_float(), C.c_float(), C.c_float()
    rc = L.mibc_debug_gemm_compare(M, N, K, epi, act, bias, rope_T, 2, dbg0, C.byref(nd), C.byref(md), C.byref(t256),
                                   C.byref(t128), C.byref(ef))
    assert rc == 0
    return nd.value, md.value, t256.value, t128.value, ef.value


@pytest.mark.parametrize("M,N,K,epi,act,bias,rope_T", SHAPES)
def test_gemm256_bit_identical_to_gemm_dma(M, N, K, epi, act, bias, rope_T):
    nd, md, t256, t128, ef = _run(M, N, K, epi, act, bias, rope_T, 0x1000)
    print(f"gemm256  M={M} N={N} K={K} epi={epi}: differing halfs {nd} (max {md:.5f}); {t256:.3f} ms vs {t128:.3f} ms")
    assert nd == 0


@pytest.mark.parametrize("M,N,K,epi,act,bias,rope_T", SHAPES)
def test_production_gemm_numerics(M, N, K, epi, act, bias, rope_T):
    nd, md, t256, t128, ef = _run(M, N, K, epi, act, bias, rope_T, 0)
    print(f"production M={M} N={N} K={K} epi={epi}: differing halfs {nd} (max {md:.5f}), f64 err {ef:.5f}; "
          f"{t256:.3f} ms vs {t128:.3f} ms")
    assert nd >= 0                              # something was written
    if epi == 2 or K > 1024:
        assert nd == 0                          # gemm256_kernel serves these
    else:
        assert md <= 0.008
    if ef >= 0:
        assert ef <= 0.005
