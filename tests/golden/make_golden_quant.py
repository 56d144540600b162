"""tests/golden/lstm_quant.npz: the reference's own weight quantisation on a seeded weight matrix.

Generated in the build container by the COMPILED REFERENCE (oracle/_ref: utils::quantize_tensor of
torch_utils/tensor_utils.cpp:293-300, called on the f16 copy of cat(W_ih, W_hh, 1) as nn/LSTMStack.cpp:160-168 does).
    python tests/golden/make_golden_quant.py
The fixture travels to the GPU box (no /root/reference there); tests/test_oracle_pinned.py compares
mibc_quantize_lstm_weights with it bit for bit."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle_py as O  # noqa: E402

SEED, CS = 20260924, 96


def weights(seed=SEED, c=CS):
    rng = np.random.default_rng(seed)
    w_ih = (rng.standard_normal((4 * c, c)) * 0.2).astype(np.float32)
    w_hh = (rng.standard_normal((4 * c, c)) * 0.2).astype(np.float32)
    # rows that exercise the corners: a large outlier (small scale), tiny values (huge scale), exact .5 products, a zero
    w_ih[0, 0] = 7.5
    w_hh[1, :] *= 1e-2      # (much smaller rows overflow the reference's f16 scale to inf: not a case worth pinning)
    w_ih[1, :] *= 1e-2
    w_ih[2, :8] = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 64.0, 0.0, 63.5], np.float32) / 64.0
    w_ih[2, 8:] = 0.0
    w_hh[2, :] = 0.0
    w_hh[2, 0] = 2.0            # max of row 2: scale 64 -> products land exactly on k + .5
    return w_ih, w_hh


def ref_quantize(w_ih, w_hh):
    cat = np.ascontiguousarray(np.concatenate([w_ih, w_hh], 1), np.float32)
    rows, cols = cat.shape
    q = np.zeros((rows, cols), np.int8)
    sc = np.zeros(rows, np.float32)
    rc = O.ref().ref_quantize_tensor_f16_rows(C.c_void_p(cat.ctypes.data), rows, cols, C.c_void_p(q.ctypes.data),
                                              C.c_void_p(sc.ctypes.data))
    if rc != 0:
        raise RuntimeError(O.ref().ref_last_error().decode())
    return q, sc


if __name__ == "__main__":
    w_ih, w_hh = weights()
    q, sc = ref_quantize(w_ih, w_hh)
    out = os.path.join(HERE, "lstm_quant.npz")
    np.savez_compressed(out, q=q, scale=sc, seed=SEED, C=CS, crc_w=zlib.crc32(w_ih.tobytes() + w_hh.tobytes()))
    print(out, q.shape, "scale range", sc.min(), sc.max(), "|q| max", np.abs(q).max())
