"""Windowed attention kernels of the transformer path (csrc/tx.hip, pytest -m gpu): the LDS-ring kernel
(window_attention_v3_kernel: one workgroup walks all query tiles of a (chunk, head) pair, a tile brings in only its 128
new keys) against the re-staging kernel it replaces (v2, still used for short / ragged sequences).  Both perform the
same operations per (query, key tile), so the contract is bit-identity; the numerics against the reference are pinned
by the transformer parity tests (test_gpu_parity.py, test_gpu_baseline_parity.py), which run the ring kernel at
T = 1024 tokens."""
import ctypes as C

import pytest

from dorado_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,T,H", [(3, 1024, 8), (2, 256, 2), (5, 640, 4)])
def test_ring_attention_bit_identical_to_restaging_kernel(N, T, H):
    L = capi.dbg_lib()
    L.mibc_debug_attention_compare.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_longlong), C.POINTER(C.c_float),
                                                               C.POINTER(C.c_float)]
    nd, t3, t2 = C.c_longlong(-1), C.c_float(), C.c_float()
    rc = L.mibc_debug_attention_compare(N, T, H, 127, 128, 2, C.byref(nd), C.byref(t3), C.byref(t2))
    print(f"N={N} T={T} H={H}: differing halfs {nd.value}; ring {t3.value:.3f} ms, re-staging {t2.value:.3f} ms")
    assert rc == 0 and nd.value == 0
