#!/usr/bin/env python3
"""No GPU: the reference's OWN BasecallerNode -> HipModelRunnerAdapter -> HipModelRunner -> HipCaller -> C-ABI test double
(oracle/_ref/libmibc_adapter_fake.so: integration/ drivers + dorado_amd/host sources + tools/fake_mibc.cpp + the reference's
BasecallerNode / MessageSink / chunk / stitch compiled in place), against this repo's node over the same double.  The double
"calls" a chunk by a context-dependent function of its samples, so chunk plans, queue choice, repeat padding, the variable-chunk
row packing (several chunks per row, first fit), the node's 32-row-span budget vs the runner's batch_size(), overflow batches and
stitching all show up in the reads.  Run in its own process (tests/test_adapter.py) so the double never meets the real libmibc.so.
    python tools/ref_node_over_fake_engine.py [n_reads]        prints one JSON line"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (libtorch first)

from dorado_amd import config, hostapi  # noqa: E402

SO = os.path.join(ROOT, "oracle", "_ref", "libmibc_adapter_fake.so")
L = C.CDLL(SO)
L.adapter_last_error.restype = C.c_char_p
L.mibch_last_error.restype = C.c_char_p
L.mibch_generate_chunks.restype = C.c_long
L.mibch_stitch_chunks.restype = C.c_long
hostapi._lib = L                      # hostapi's wrappers now talk to the host layer inside the double's library

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(123)
res = {}
for variable in (0, 1):
    cfg = config.tiny(256, 4)
    cfg.lstm_layers = 3
    cfg.chunk_size, cfg.overlap = 1200, 120
    cfg.qscale, cfg.qbias = 1.0, 0.0
    cfg.normalise_basecaller_params()
    ws = [np.zeros(4, np.float32)]
    # many short reads (several chunks per row in variable mode), reads around the two chunk sizes, long reads
    lens = np.concatenate([rng.integers(6, 400, n_reads // 3), rng.integers(400, 2600, n_reads // 3),
                           rng.integers(2600, 15000, n_reads - 2 * (n_reads // 3)), [600, 606, 1194, 1200, 1206, 2280, 66, 6]])
    rng.shuffle(lens)
    # a run of reads just over half a row: one per row, so the node's step budget (0.8 x rows x (T + 2)) admits more chunks than
    # there are rows -> HipModelRunner's overflow batch
    lens = np.concatenate([lens[:200], rng.integers(606, 700, 400), lens[200:]])
    reads = [rng.integers(0, 65535, int(n)).astype(np.uint16).view(np.float16) for n in lens]
    d = cfg.to_desc()
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    numel = (C.c_int64 * len(ws))(*[w.size for w in ws])
    n = len(reads)
    pitch = int(max(lens)) // cfg.stride + 8
    seq, qs, mv = (np.zeros((n, pitch), np.uint8) for _ in range(3))
    sl, ml = np.zeros(n, np.int64), np.zeros(n, np.int64)
    st = (C.c_double * 8)()
    sig = np.ascontiguousarray(np.concatenate(reads))
    rl = np.array(lens, np.int64)
    RESTART_AFTER = n // 2             # NodeSmokeTest.cpp's restart case: terminate + restart of node and runners half way
    batch = 128 if variable else 64     # 128 rows: batch_size() = 96 rows' worth of steps > 128 one-chunk rows of such reads
    rc = L.adapter_run_basecaller_node(C.byref(d), arr, numel, len(ws), b"hip:0", 2, cfg.chunk_size, cfg.overlap, batch, variable,
                                       C.c_float(cfg.qscale), C.c_float(cfg.qbias), sig.ctypes.data_as(C.c_void_p),
                                       rl.ctypes.data_as(C.c_void_p), n, pitch, seq.ctypes.data_as(C.c_void_p),
                                       qs.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p),
                                       sl.ctypes.data_as(C.c_void_p), ml.ctypes.data_as(C.c_void_p), st, RESTART_AFTER)
    if rc != 0:
        print(json.dumps({"error": L.adapter_last_error().decode(), "variable": variable}))
        sys.exit(1)
    if variable:
        want, hst = hostapi.basecall_reads(cfg, ws, reads, device="hip:0", num_runners=2, batch_size=batch, variable_chunks=True)
    else:
        want, hst = hostapi.basecall_reads(cfg, ws, reads, device="hip:0", num_runners=2, batch_size=batch, two_queues=True)
    bad = [r for r in range(n) if seq[r, :sl[r]].tobytes().decode() != want[r][0] or qs[r, :sl[r]].tobytes().decode() != want[r][1]
           or ml[r] != len(want[r][2]) or (mv[r, :ml[r]] != want[r][2]).any() or ml[r] != lens[r] // cfg.stride]
    res["variable" if variable else "fixed"] = {
        "reads": n, "differing_reads": len(bad), "first_differing": (int(bad[0]), int(lens[bad[0]])) if bad else None,
        "bases": int(sl.sum()), "runners_variable": int(st[4]), "ref_node_batches": st[0] + st[1],
        "ref_node_samples_processed": st[2], "host_node_batches": hst["batches_called"],
        "var_engine_batches": st[5], "var_overflow_batches": st[6]}
print(json.dumps(res))
