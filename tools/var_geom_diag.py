#!/usr/bin/env python3
"""Diagnostic (GPU): a 300-sample variable chunk must be called identically whatever the batch geometry / its neighbours.
Prints the calls of the short reads through: the C-ABI engine alone at T_in = 600 and 1200, the host node with one geometry
(chunk size 1200 and 600), and the reference's BasecallerNode over the adapter (two geometries) with all reads / short reads only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, hostapi, synth  # noqa: E402

cfg = config.tiny(128, 4)
cfg.lstm_layers = 5
cfg.chunk_size, cfg.overlap = 1200, 120
cfg.qscale, cfg.qbias = 1.05, -0.3
cfg.normalise_basecaller_params()
ws = [np.ascontiguousarray(w, np.float32) for w in synth.make_weights(cfg, seed=17)]
lens = [300, 594, 600, 1200, 1206, 2500, 3343, 5010, 809, 4106, 7777, 12000, 312]
reads = [synth.make_signal(1, L_, seed=300 + i)[0] for i, L_ in enumerate(lens)]
short = [0, 1, 12]


def show(tag, seqs):
    print(f"{tag:58s}", " | ".join(s[-12:] for s in seqs), flush=True)


# A: the engine alone, one chunk per call, at two geometries
for t_in in (600, 1200):
    eng = capi.Engine(cfg, ws)
    out = []
    for r in short:
        x = np.zeros((64, t_in), np.float16)
        x[0, :lens[r]] = reads[r]
        out.append(eng.call_var(x, [(0, 0, lens[r])])[0][0])
    show(f"engine alone, T_in {t_in}", out)
    # same chunk with garbage behind it in the row and a neighbour chunk
    out = []
    for r in short:
        x = np.full((64, t_in), 3.0, np.float16)
        x[0, :lens[r]] = reads[r]
        nb = min(150, t_in - lens[r] - 12) // 6 * 6
        ch = [(0, 0, lens[r])] + ([(0, lens[r] + 12, nb)] if nb >= 60 else [])
        out.append(eng.call_var(x, ch)[0][0])
    show(f"engine, garbage + neighbour in the row, T_in {t_in}", out)
    eng.close()

# B: host node, one geometry
for cs in (1200, 600):
    c2 = config.tiny(128, 4)
    c2.lstm_layers = 5
    c2.chunk_size, c2.overlap = cs, 120
    c2.qscale, c2.qbias = 1.05, -0.3
    c2.normalise_basecaller_params()
    want, _ = hostapi.basecall_reads(c2, ws, reads, device="hip:0", num_runners=2, batch_size=64, variable_chunks=True)
    show(f"host node variable, chunk size {cs}, all reads", [want[r][0] for r in short])
    want, _ = hostapi.basecall_reads(c2, ws, [reads[r] for r in short], device="hip:0", num_runners=2, batch_size=64, variable_chunks=True)
    show(f"host node variable, chunk size {cs}, short reads only", [w[0] for w in want])

# C: the reference's BasecallerNode over the adapter
import torch  # noqa: E402,F401
capi.lib()
C.CDLL(hostapi.LIB_PATH, mode=C.RTLD_GLOBAL)
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmibc_adapter_test.so"))
L.adapter_last_error.restype = C.c_char_p


RESTART_AFTER = 0


def node(idx, variable=1):
    rr = [reads[i] for i in idx]
    ll = [lens[i] for i in idx]
    d = cfg.to_desc()
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    numel = (C.c_int64 * len(ws))(*[w.size for w in ws])
    n = len(rr)
    pitch = max(ll) // cfg.stride + 8
    seq, qs, mv = (np.zeros((n, pitch), np.uint8) for _ in range(3))
    sl, ml = np.zeros(n, np.int64), np.zeros(n, np.int64)
    st = (C.c_double * 8)()
    sig = np.ascontiguousarray(np.concatenate(rr).astype(np.float16))
    rl = np.array(ll, np.int64)
    rc = L.adapter_run_basecaller_node(C.byref(d), arr, numel, len(ws), b"hip:0", 2, cfg.chunk_size, cfg.overlap, 64, variable,
                                       C.c_float(cfg.qscale), C.c_float(cfg.qbias), sig.ctypes.data_as(C.c_void_p),
                                       rl.ctypes.data_as(C.c_void_p), n, pitch, seq.ctypes.data_as(C.c_void_p),
                                       qs.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p),
                                       sl.ctypes.data_as(C.c_void_p), ml.ctypes.data_as(C.c_void_p), st, RESTART_AFTER)
    assert rc == 0, L.adapter_last_error().decode()
    return [seq[r, :sl[r]].tobytes().decode() for r in range(n)]


allr = node(list(range(len(lens))))
show("reference node + adapter, variable, all reads", [allr[r] for r in short])
show("reference node + adapter, variable, short reads only", node(short))
show("reference node + adapter, variable, short reads only (again)", node(short))
