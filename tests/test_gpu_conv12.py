"""conv1 + conv2 front end (csrc/conv.hip).  Round 5 moved conv2's accumulation from v_fma_f32 onto the f32-input matrix
instruction v_mfma_f32_16x16x4_f32, whose result is a k-ordered fmaf chain from C — the same order as the VALU loop it replaced.
Contract: EVERY output bit unchanged — the scores of the whole network through the product library (MFMA conv2) equal, byte for
byte, the scores through the debug library with MIBC_CONV12_VALU=1 (the round 1-4 loop), for f16 and raw-int16 input, fixed
and variable chunk sizes, both activation pairs.  (pytest -m gpu; the debug run is a child process: the switch is read once.)"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _digests():
    """sha256 of the scores of four small runs (this process's library)."""
    from dorado_amd import capi, config, synth
    out = {}
    for name, C, acts in (("tanh128", 128, None), ("clamp256", 256, (1, 1))):
        cfg = config.tiny(C, 4)
        cfg.lstm_layers = 2
        if acts:
            cfg.convs[0].activation, cfg.convs[1].activation = acts
        ws = synth.make_weights(cfg, seed=5)
        eng = capi.Engine(cfg, ws)
        x = synth.make_signal(64, 1203 * 2, seed=6)            # not a multiple of the 256-step workgroup tile
        out[name + "_f16"] = hashlib.sha256(np.ascontiguousarray(eng.forward(x)).tobytes()).hexdigest()
        rng = np.random.default_rng(7)
        raw = (480 + 95 * x.astype(np.float32)).astype(np.int16)
        ss = np.stack([rng.uniform(400, 560, 64), rng.uniform(60, 120, 64)], 1).astype(np.float32)
        X = np.zeros((64, 2400), np.int16)
        X[:, :2400] = raw[:, :2400]
        chunks = [(r, 0, 600 + 6 * r) for r in range(64)] + [(r, 1200, 1200) for r in range(0, 64, 3)]
        chunks.sort()
        S = eng.forward_var(X, chunks, ss)                      # (the gaps between chunks hold garbage: hash the chunks only)
        st = cfg.stride
        h = hashlib.sha256()
        for r, s0, L in chunks:
            h.update(np.ascontiguousarray(S[r, s0 // st:(s0 + L) // st]).tobytes())
        out[name + "_var_i16"] = h.hexdigest()
        eng.close()
    return out


def test_mfma_conv2_is_bit_identical_to_the_valu_loop():
    here = _digests()
    env = dict(os.environ, MIBC_CONV12_VALU="1", MIBC_TEST_LIB="dbg")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-800:]
    there = json.loads(r.stdout.strip().splitlines()[-1])
    assert here == there, (here, there)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from dorado_amd import capi
    if os.environ.get("MIBC_TEST_LIB") == "dbg":
        capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libmibc_dbg.so")
    print(json.dumps(_digests()))
