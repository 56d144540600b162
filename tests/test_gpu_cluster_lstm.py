"""Cluster (hidden-split) LSTM kernel, lstm_cluster.hip (pytest -m gpu).

The cluster kernel (C = 512 / 768 / 1024, batch a multiple of 256 rows) performs, element for element, the
arithmetic of the per-workgroup kernel it replaces (same MFMA shape, same k order, same gate functions), so
the contract is BIT-IDENTITY of the LSTM stack's output between
    a batch of N = 256 k rows        -> cluster kernel (KCL = C/128 workgroups exchange h every step), and
    the same rows in batches that are not a multiple of 256 -> lstm_layer_xg_kernel (no exchange),
which the BASELINE-size parity test pins to the reference (test_gpu_baseline_parity.py, N = 32).
The small grids here put the members of a cluster on DIFFERENT XCDs (block b -> XCD b % 8), so the hand-off
protocol is exercised across non-coherent L2s; the large case uses the same-XCD mapping, several clusters per
XCD and more clusters than fit at once (row-group loop)."""
import numpy as np
import pytest

from dorado_amd import capi, config, synth

pytestmark = pytest.mark.gpu


def _cfg(C, state_len=3, layers=5):
    cfg = config.tiny(C, state_len)
    cfg.lstm_layers = layers
    return cfg


def _lstm_out(eng, x16):
    """LSTM stack output [T][N][C] f16 of a forward call (parity tap 3) and the scores."""
    sc = eng.forward(x16)
    T = eng.output_steps(x16.shape[1])
    return eng.tap(3, (T, x16.shape[0], eng.cfg.lstm_size), np.float16), sc


@pytest.mark.parametrize("C,T_in", [(1024, 606), (512, 906), (768, 606)])
def test_cluster_kernel_bit_identical_to_per_workgroup_kernel(C, T_in):
    cfg = _cfg(C)
    ws = synth.make_weights(cfg, seed=80 + C)
    x = synth.make_signal(256, T_in, seed=81)
    eng = capi.Engine(cfg, ws)
    g = eng.batch_granularity()
    a_cl, s_cl = _lstm_out(eng, x)                       # 256 rows: one cluster
    # same rows through the per-workgroup kernel: 256 - g and g rows (neither is a multiple of 256)
    a1, s1 = _lstm_out(eng, x[: 256 - g])
    a2, s2 = _lstm_out(eng, x[256 - g:])
    a_wg = np.concatenate([a1, a2], axis=1)
    s_wg = np.concatenate([s1, s2], axis=0)
    nbad = int((a_cl.view(np.uint16) != a_wg.view(np.uint16)).sum())
    d = np.abs(a_cl.astype(np.float32) - a_wg.astype(np.float32))
    print(f"C={C}: LSTM output elements differing {nbad} of {a_cl.size}, max-abs {d.max():.5f}")
    assert np.isfinite(a_cl.astype(np.float32)).all()
    assert nbad == 0, f"cluster kernel differs from the per-workgroup kernel in {nbad} elements (max {d.max()})"
    assert (s_cl.view(np.uint16) == s_wg.view(np.uint16)).all()
    # determinism across repeated launches (hand-off races would show up as run-to-run differences)
    for _ in range(3):
        b, _ = _lstm_out(eng, x)
        assert (b.view(np.uint16) == a_cl.view(np.uint16)).all()
    eng.close()


def test_cluster_kernel_many_clusters_and_row_groups():
    """N = 40 x 256 rows at C = 1024: 32 clusters resident (4 per XCD), 8 more in a second round of the
    row-group loop.  The batch tiles 256 distinct rows, so every cluster must reproduce cluster 0, and cluster 0
    must equal the per-workgroup kernel on those rows."""
    cfg = _cfg(1024, 3, 3)
    ws = synth.make_weights(cfg, seed=90)
    T_in = 246
    base = synth.make_signal(256, T_in, seed=91)
    eng = capi.Engine(cfg, ws)
    big = np.tile(base, (40, 1))
    a_big, _ = _lstm_out(eng, big)
    a_wg1, _ = _lstm_out(eng, base[:224])
    a_wg2, _ = _lstm_out(eng, base[224:])
    a_wg = np.concatenate([a_wg1, a_wg2], axis=1)
    T = a_big.shape[0]
    tiles = a_big.reshape(T, 40, 256, 1024)
    for k in range(40):
        assert (tiles[:, k].view(np.uint16) == a_wg.view(np.uint16)).all(), f"cluster {k} differs"
    eng.close()


def test_cluster_kernel_in_the_full_call_path():
    """sup@v4.3 shape through mibc_call at N = 256 (cluster kernel) == N = 224 + 32 (per-workgroup kernel):
    identical calls."""
    cfg = config.sup_v43()
    ws = synth.make_weights(cfg, seed=52)
    x = synth.make_signal(256, 1206, seed=53)
    eng = capi.Engine(cfg, ws)
    got = eng.call(x)
    want = eng.call(x[:224]) + eng.call(x[224:])
    for a, b in zip(got, want):
        assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all()
    assert sum(len(a[0]) for a in got) > 1000
    eng.close()
