"""Windowed attention kernels of the transformer path (csrc/tx.hip, pytest -m gpu): the LDS-ring kernel
(window_attention_v3_kernel: one workgroup walks all query tiles of a (chunk, head) pair, a tile brings in only its 128
new keys; round 4: masks on the two boundary key tiles only, exp2 with scale and row maximum folded into one packed fma,
1 / sum applied to the outputs) and the re-staging kernel (v2, used for short / ragged sequences) against a host f64
restatement of the reference's windowed scaled-dot-product attention (nn/TxModules.cpp:398-418, incl. the CPU path's
12-split slice).  Stated tolerance: max-abs <= 2e-3 on outputs in [-1, 1] (two f16 ulps at 1.0: probabilities and outputs
are rounded to f16).  The whole-model numerics are pinned by the transformer parity tests (test_gpu_parity.py,
test_gpu_baseline_parity.py), which run the ring kernel at T = 1024 tokens."""
import ctypes as C

import pytest

from dorado_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,T,H", [(3, 1024, 8), (2, 256, 2), (5, 640, 4)])
def test_attention_kernels_vs_host_reference(N, T, H):
    L = capi.dbg_lib()
    L.mibc_debug_attention_compare.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_longlong)] + [C.POINTER(C.c_float)] * 4
    nd, t3, t2, e3, e2 = C.c_longlong(-1), C.c_float(), C.c_float(), C.c_float(-1), C.c_float(-1)
    rc = L.mibc_debug_attention_compare(N, T, H, 127, 128, 2, C.byref(nd), C.byref(t3), C.byref(t2), C.byref(e3), C.byref(e2))
    print(f"N={N} T={T} H={H}: ring {t3.value:.3f} ms (max err {e3.value:.2e}), re-staging {t2.value:.3f} ms (max err "
          f"{e2.value:.2e}); halfs differing between the two: {nd.value}")
    assert rc == 0
    assert 0 <= e3.value <= 2e-3 and 0 <= e2.value <= 2e-3
