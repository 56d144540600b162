"""The quantised LSTM path (csrc/lstm_q8.hip, mibc_model_desc::lstm_quant — the reference's KOI_I8 path,
nn/LSTMStack.cpp:127-211): int8 weights with per-row scales, int8 activations, f16 first layer (pytest -m gpu).

An 8-bit path has its own, STATED tolerance (the f16 path stays the parity headline):
  * against the f16 path of the same engine on the same weights (small models here): scores rms <= 0.15 (round(127 h)
    activations: step 0.0079, rms error 0.0023 per element, amplified by the random-weight layers), deterministic;
  * against the COMPILED REFERENCE at BASELINE size: test_gpu_baseline_parity.py::test_quantised_lstm_vs_reference
    (scores rms / max and identity on the reference's confident bases; measured values in DESIGN.md)."""
import os

import numpy as np
import pytest

from dorado_amd import capi, config, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,state_len", [(128, 4), (256, 4), (384, 4)])
def test_quantised_path_runs_and_tracks_f16_path(C, state_len):
    cfg = config.tiny(C, state_len)
    cfg.lstm_layers = 5
    ws = synth.make_weights(cfg, seed=11)
    x = synth.make_signal(128, 1206, seed=12)
    e16 = capi.Engine(cfg, ws)
    s16 = e16.forward(x).astype(np.float32)
    c16 = e16.call(x)
    e16.close()
    cfg.lstm_quant = True
    e8 = capi.Engine(cfg, ws)
    s8 = e8.forward(x).astype(np.float32)
    c8 = e8.call(x)
    # deterministic: the same call twice is bit-identical
    assert (e8.forward(x).astype(np.float32) == s8).all()
    e8.close()
    d = np.clip(s8, -5, 5) - np.clip(s16, -5, 5)
    rms, mx = float(np.sqrt((d ** 2).mean())), float(np.abs(d).max())
    from parity_utils import identity
    ids = [identity(a[0], b[0]) for a, b in zip(c8, c16)]
    print(f"C={C}: int8 path vs f16 path: scores rms {rms:.4f} max {mx:.3f}; identity median {np.median(ids):.3f}")
    assert np.isfinite(s8).all()
    assert rms <= 0.15, rms
    assert np.median(ids) >= 0.85
    # ... and against the f32 ORACLE (oracle.c restatement of the reference's CPU path; no oracle restates Koi's int8
    # activation arithmetic — closed source — so the 8-bit path is held to the f32 network with its own tolerance)
    from oracle import oracle_py as O
    s_o = O.lstm_crf_forward(cfg, ws, x[:8].astype(np.float32)[:, None, :])
    do = np.clip(s8[:8], -5, 5) - s_o
    rms_o = float(np.sqrt((do ** 2).mean()))
    d16 = np.clip(s16[:8], -5, 5) - s_o
    print(f"C={C}: int8 path vs f32 oracle: scores rms {rms_o:.4f} (f16 path: {float(np.sqrt((d16 ** 2).mean())):.4f})")
    assert rms_o <= 0.15, rms_o


def test_quant_rejects_unsupported_shapes():
    cfg = config.tiny(96, 3)
    cfg.lstm_quant = True
    with pytest.raises(capi.MibcNotSupported):
        capi.Engine(cfg, synth.make_weights(cfg, seed=1))
    # the wide (cluster) instance needs whole 256-row clusters: the engine says so through its batch granularity (round 5,
    # HipCaller::choose_batch_size rounds requests and memory caps to it) and refuses anything else as an argument error
    cfg = config.tiny(512, 5)
    cfg.lstm_quant = True
    eng = capi.Engine(cfg, synth.make_weights(cfg, seed=1))
    assert eng.batch_granularity() == 256
    with pytest.raises(capi.MibcError):
        eng.forward(synth.make_signal(64, 306, seed=2))
    eng.close()
    cfg.lstm_quant = False
    eng = capi.Engine(cfg, synth.make_weights(cfg, seed=1))
    assert eng.batch_granularity() == 64
    eng.close()


@pytest.mark.parametrize("C,state_len,layers", [(512, 5, 3), (768, 5, 2), (1024, 5, 5)])
def test_quantised_cluster_path_tracks_f16_path_and_oracle(C, state_len, layers):
    """Round 4: the int8 instance of the CU-cluster kernel (csrc/lstm_cluster.hip, Q8; lstm_size 512 / 768 / 1024): layer 0 in
    f16 + conversion, the middle layers int8 -> int8, the last layer int8 -> f16 with an int8 exchange copy.  Same stated
    tolerance as the narrow int8 kernel: scores rms <= 0.15 against the f16 path and against the f32 oracle; deterministic."""
    cfg = config.tiny(C, state_len)
    cfg.lstm_layers = layers
    ws = synth.make_weights(cfg, seed=21)
    x = synth.make_signal(512, 606, seed=22)       # two clusters of 256 rows
    e16 = capi.Engine(cfg, ws)
    s16 = e16.forward(x).astype(np.float32)
    c16 = e16.call(x)
    e16.close()
    cfg.lstm_quant = True
    e8 = capi.Engine(cfg, ws)
    s8 = e8.forward(x).astype(np.float32)
    c8 = e8.call(x)
    assert (e8.forward(x).astype(np.float32) == s8).all()
    e8.close()
    d = np.clip(s8, -5, 5) - np.clip(s16, -5, 5)
    rms = float(np.sqrt((d ** 2).mean()))
    from parity_utils import identity
    from oracle import oracle_py as O
    ids = [identity(a[0], b[0]) for a, b in zip(c8, c16)]
    s_o = O.lstm_crf_forward(cfg, ws, x[:4].astype(np.float32)[:, None, :])
    rms_o = float(np.sqrt(((np.clip(s8[:4], -5, 5) - s_o) ** 2).mean()))
    print(f"C={C}: int8 cluster path vs f16 path: scores rms {rms:.4f} max {np.abs(d).max():.3f}; identity median "
          f"{np.median(ids):.3f}; vs f32 oracle rms {rms_o:.4f}")
    assert np.isfinite(s8).all()
    assert rms <= 0.15 and rms_o <= 0.15, (rms, rms_o)
    assert np.median(ids) >= 0.85


@pytest.mark.parametrize("C,state_len,N", [(256, 4, 128), (512, 5, 256)])
def test_quantised_path_with_swish_front_end_keeps_first_layer_f16(C, state_len, N):
    """The reference hands int8 to the FIRST LSTM layer only when the last convolution ends in tanh (nn/ConvStack.cpp:72); with a
    swish front end the first layer runs in f16 and its output is converted (nn/LSTMStack.cpp:199-207).  Both branches exist
    here (engine.hip q_all); the tanh models of the other tests take the all-int8 one, this test the other."""
    cfg = config.tiny(C, state_len)
    cfg.lstm_layers = 3
    cfg.convs[2].activation = config.ACT_SWISH_CLAMP
    ws = synth.make_weights(cfg, seed=31)
    x = synth.make_signal(N, 606, seed=32)
    e16 = capi.Engine(cfg, ws)
    s16 = e16.forward(x).astype(np.float32)
    e16.close()
    cfg.lstm_quant = True
    e8 = capi.Engine(cfg, ws)
    s8 = e8.forward(x).astype(np.float32)
    e8.close()
    d = np.clip(s8, -5, 5) - np.clip(s16, -5, 5)
    rms = float(np.sqrt((d ** 2).mean()))
    print(f"C={C} swish front end: int8 (layers 2..L) vs f16: scores rms {rms:.4f} max {np.abs(d).max():.3f}")
    assert np.isfinite(s8).all() and rms <= 0.15


_FUSE_SCRIPT = r"""
import hashlib, os, sys
import os

import numpy as np
sys.path.insert(0, sys.argv[1])
from dorado_amd import capi, config, synth
if sys.argv[2] == "dbg":
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmibc_dbg.so")
C_, sl, N = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cfg = config.tiny(C_, sl)
cfg.lstm_quant = True
eng = capi.Engine(cfg, synth.make_weights(cfg, seed=41))
s = eng.forward(synth.make_signal(N, 1206, seed=42))
print(hashlib.sha256(np.ascontiguousarray(s).tobytes()).hexdigest())
"""


@pytest.mark.parametrize("C,state_len,N", [(384, 4, 128), (256, 4, 64), (1024, 5, 256)])
def test_conv3_int8_epilogue_equals_the_separate_conversion_pass(C, state_len, N):
    """Round 6: with all LSTM layers int8 (tanh conv3, 128 < lstm_size) conv3's GEMM epilogue writes the
    round(127 f16(tanh)) rows itself (wsgemm_kernel<.., 4>) instead of an f16 tensor + q8_convert_kernel — what the reference
    does in two passes (nn/ConvStack.cpp:243,324-329).  Bit-identical by construction; asserted: the debug library with the
    fusion switched off (MIBC_FUSE_Q8=0), the debug library with it on and the product library give the same scores."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for lib, fuse in (("dbg", "0"), ("dbg", "1"), ("product", "1")):
        env = dict(os.environ, MIBC_FUSE_Q8=fuse)
        r = subprocess.run([sys.executable, "-c", _FUSE_SCRIPT, root, lib, str(C), str(state_len), str(N)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(r.stdout.strip().splitlines()[-1])
    assert out[0] == out[1] == out[2], out
