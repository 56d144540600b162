"""Weight-stationary cluster LSTM kernel, lstm_ws.hip (pytest -m gpu).

Enabled here through mibc_debug_set_ws_min_rows when the engine's default threshold does not select it.  For lstm_size 384 (hac) six workgroups of a cluster keep the layer's weights in
their register files and exchange h through the layer output.  It performs, element for element, the arithmetic of lstm_layer_x8_kernel (same MFMA shape, same k order, same gate functions), so the contract
is BIT-IDENTITY between a large batch (cluster kernel) and the same rows in batches below the threshold (x8), which
the BASELINE-size parity test pins to the reference (test_gpu_baseline_parity.py, N = 64)."""
import numpy as np
import pytest

from dorado_amd import capi, config, synth

pytestmark = pytest.mark.gpu


def _cfg(layers=5):
    cfg = config.tiny(384, 3)
    cfg.lstm_layers = layers
    return cfg


def _lstm_out(eng, x16):
    sc = eng.forward(x16)
    T = eng.output_steps(x16.shape[1])
    return eng.tap(3, (T, x16.shape[0], eng.cfg.lstm_size), np.float16), sc


@pytest.mark.parametrize("N,T_in", [(2048, 606), (2048 + 64 * 3, 306)])
def test_ws_kernel_bit_identical_to_x8(N, T_in):
    """12 clusters (linear map: the members of a cluster sit on different XCDs), uneven row tiles per cluster."""
    cfg = _cfg()
    ws = synth.make_weights(cfg, seed=384)
    x = synth.make_signal(N, T_in, seed=385)
    eng = capi.Engine(cfg, ws)
    eng.set_ws_min_rows(2048)
    a_ws, s_ws = _lstm_out(eng, x)
    parts = [_lstm_out(eng, x[i:i + 1024]) for i in range(0, N, 1024)]     # below the threshold: x8
    a_x8 = np.concatenate([p[0] for p in parts], axis=1)
    s_x8 = np.concatenate([p[1] for p in parts], axis=0)
    nbad = int((a_ws.view(np.uint16) != a_x8.view(np.uint16)).sum())
    d = np.abs(a_ws.astype(np.float32) - a_x8.astype(np.float32))
    print(f"N={N}: LSTM output elements differing {nbad} of {a_ws.size}, max-abs {d.max():.5f}")
    assert np.isfinite(a_ws.astype(np.float32)).all()
    assert nbad == 0
    assert (s_ws.view(np.uint16) == s_x8.view(np.uint16)).all()
    for _ in range(3):      # hand-off races would show up as run-to-run differences
        b, _ = _lstm_out(eng, x)
        assert (b.view(np.uint16) == a_ws.view(np.uint16)).all()
    eng.close()


def test_ws_kernel_full_grid():
    """N = 16384 rows: 40 clusters, 5 per XCD (same-XCD mapping), 25-26 row tiles each.  The batch tiles 256 distinct
    rows, so every 256-row block must reproduce block 0, which must equal the x8 kernel on those rows."""
    cfg = _cfg(3)
    ws = synth.make_weights(cfg, seed=90)
    T_in = 246
    base = synth.make_signal(256, T_in, seed=91)
    eng = capi.Engine(cfg, ws)
    eng.set_ws_min_rows(2048)
    a_big, _ = _lstm_out(eng, np.tile(base, (64, 1)))
    a_x8, _ = _lstm_out(eng, base)
    T = a_big.shape[0]
    tiles = a_big.reshape(T, 64, 256, 384)
    for k in range(64):
        assert (tiles[:, k].view(np.uint16) == a_x8.view(np.uint16)).all(), f"block {k} differs"
    eng.close()
