"""One process, several engines / batch dimensions (pytest -m gpu) — the single-process side of the multi-GPU design
(api/runner_creation.cpp:85-124: one caller per device, callers built on concurrent threads; CudaCaller.cpp:204-214:
one FIFO per GPU; :382-413: several batch dimensions per caller).  On the 1-GPU box every engine sits on device 0; the
per-device launch state (csrc/common.h MIBC_LDS_ATTR_ONCE / mibc_ncu), the per-device cluster gate and the shared device
queue are exactly what N devices use."""
import threading

import numpy as np
import pytest

from dorado_amd import capi, config, hostapi, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _cfg(C, state_len, layers=3):
    cfg = config.tiny(C, state_len)
    cfg.lstm_layers = layers
    return cfg


def _two_engines_concurrently(C, state_len, N, t_in, devices):
    cfg = _cfg(C, state_len)
    ws = [synth.make_weights(cfg, seed=10 + i) for i in range(2)]
    xs = [synth.make_signal(N, t_in, seed=20 + i) for i in range(2)]
    want = []
    for w, x, dev in zip(ws, xs, devices):
        e = capi.Engine(cfg, w, device=dev)
        want.append(e.call(x))
        e.close()
    got = [None, None]
    errs = []
    barrier = threading.Barrier(2)

    def work(i):
        try:
            barrier.wait()
            e = capi.Engine(cfg, ws[i], device=devices[i])   # concurrent mibc_create: weight uploads, attribute set-up
            barrier.wait()
            for _ in range(4):                   # concurrent launches on two streams of one device
                got[i] = e.call(xs[i])
                for a, b in zip(got[i], want[i]):
                    assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all()
            e.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("C,state_len,N,t_in", [(128, 4, 128, 606), (512, 5, 512, 306)])
def test_engines_created_and_run_concurrently_from_two_threads(C, state_len, N, t_in):
    """Two threads each create an engine and call it in a loop (C = 512: the CU-cluster LSTM kernel, whose launches
    from different streams must not overlap on one device — cluster_util.h MibcClusterLaunch).  Results must equal the
    sequential ones, call after call."""
    _two_engines_concurrently(C, state_len, N, t_in, (0, 0))


@pytest.mark.parametrize("C,state_len,N,t_in", [(128, 4, 128, 606), (512, 5, 512, 306)])
def test_engines_on_two_devices_concurrently(C, state_len, N, t_in):
    """The same on devices 0 AND 1 (skipped on a 1-GPU box): the first exercise of the per-device state with dev != 0 —
    MIBC_LDS_ATTR_ONCE's per-device bit (the > 64 KB LDS opt-in is a per-device function attribute), mibc_ncu()'s per-device
    cache and the per-device cluster gate (two cluster launches on DIFFERENT devices must not serialise on each other)."""
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs")
    _two_engines_concurrently(C, state_len, N, t_in, (0, 1))


def test_single_process_host_layer_over_all_devices():
    """hip:all through the C++ host layer (one HipCaller per device, shared chunk queues): calls must equal device 0's alone
    (skipped on a 1-GPU box)."""
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs")
    cfg = _cfg(128, 4)
    ws = synth.make_weights(cfg, seed=3)
    reads = [synth.make_signal(1, 3000 + 37 * i, seed=40 + i)[0] for i in range(48)]
    want, _ = hostapi.basecall_reads(cfg, ws, reads, device="hip:0", num_runners=2, batch_size=64)
    got, _ = hostapi.basecall_reads(cfg, ws, reads, device="hip:all", num_runners=2, batch_size=64)
    assert [g[0] for g in got] == [w[0] for w in want] and [g[1] for g in got] == [w[1] for w in want]


@pytest.mark.parametrize("model", ["lstm", "tx"])
def test_one_engine_serves_two_chunk_sizes_bit_identically(model):
    """mibc_reserve for the largest chunk size, then calls alternate between chunk sizes inside the same workspace
    (the engine re-derives its geometry and re-zeroes the padding rows in stream order): every call must equal a
    fresh engine's, bit for bit."""
    if model == "lstm":
        cfg = _cfg(256, 4)
        sizes, N = [1200, 600], 128
    else:
        cfg = config.tiny_tx()
        g = 16 * cfg.conv_stride
        sizes, N = [g * 16, g * 8], 16     # 256 and 128 tokens: the transposed-V attention path on both
    ws = synth.make_weights(cfg, seed=5)
    xs = {t: synth.make_signal(N, t, seed=t) for t in sizes}
    want = {}
    for t in sizes:
        e = capi.Engine(cfg, ws)
        want[t] = (e.forward(xs[t]), e.call(xs[t]))
        e.close()
    e = capi.Engine(cfg, ws)
    e.reserve(N, sizes[0])
    free0, _ = capi.device_memory(0)
    for t in [sizes[0], sizes[1], sizes[1], sizes[0], sizes[1]]:
        sc, calls = e.forward(xs[t]), e.call(xs[t])
        assert (sc.view(np.uint16) == want[t][0].view(np.uint16)).all(), f"scores differ at chunk size {t}"
        for a, b in zip(calls, want[t][1]):
            assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all()
    free1, _ = capi.device_memory(0)
    assert abs(free1 - free0) < (64 << 20), "switching the chunk size must not re-allocate the workspace"
    e.close()


def test_host_layer_two_queues_one_caller_wide_model():
    """The 0.5x chunk-size queue on a C = 512 model (cluster LSTM kernel): ONE HipCaller per device serves both batch
    dimensions through the device FIFO — stitched reads == oracle stitch of the engine's per-chunk calls."""
    cfg = _cfg(512, 4)
    cfg.chunk_size, cfg.overlap = 1200, 120
    cfg.normalise_basecaller_params()
    ws = synth.make_weights(cfg, seed=41)
    sizes = hostapi.simplex_chunk_sizes(cfg, cfg.chunk_size, cfg.overlap)
    assert sizes == [1200, 600]
    rng = np.random.default_rng(1)
    lens = [int(v) for v in rng.integers(200, 4000, size=300)]
    reads = [synth.make_signal(1, L, seed=300 + i)[0] for i, L in enumerate(lens)]
    got, stats = hostapi.basecall_reads(cfg, ws, reads, device="hip:0", num_runners=2, batch_size=256, two_queues=True)
    assert stats["samples_processed"] == sum(lens) and stats["batches_called"] >= 3
    eng = capi.Engine(cfg, ws)
    for cs in sizes:
        chunks, owner = [], []
        for r, sig in enumerate(reads):
            if sizes[hostapi.get_chunk_queue_idx(sizes, len(sig))] != cs:
                continue
            for o in O.generate_chunks(len(sig), cs, cfg.stride, cfg.overlap):
                sl = sig[o:o + cs]
                if len(sl) != cs:
                    n, ov = divmod(cs, len(sl))
                    sl = np.concatenate([np.tile(sl, n), sl[:ov]])
                chunks.append(sl)
                owner.append((r, o))
        calls = []
        for i in range(0, len(chunks), 256):
            x = np.zeros((256, cs), np.float16)
            part = chunks[i:i + 256]
            x[:len(part)] = np.stack(part)
            calls += eng.call(x)[:len(part)]
        for r in sorted({rr for rr, _ in owner}):
            idx = [i for i, (rr, _) in enumerate(owner) if rr == r]
            st = O.stitch_chunks([owner[i][1] for i in idx], [cs] * len(idx), [calls[i][2] for i in idx],
                                 [calls[i][0] for i in idx], [calls[i][1] for i in idx], len(reads[r]), cfg.stride)
            assert got[r][0] == st[0] and got[r][1] == st[1] and (got[r][2] == st[2]).all(), f"read {r} (chunk size {cs})"
    eng.close()


def test_variable_chunks_beyond_one_decode_sub_batch():
    """K = 4096 models decode in sub-batches of 4096 rows; a variable-chunk batch larger than that is decoded sub-batch
    by sub-batch (chunk table ordered by sub-batch inside the engine): every chunk == the chunk called alone."""
    cfg = _cfg(512, 5, layers=2)
    ws = synth.make_weights(cfg, seed=8)
    stride, t_in, N = cfg.stride, 60 * cfg.stride, 4096 + 256
    rng = np.random.default_rng(2)
    x = synth.make_signal(N, t_in, seed=9)
    table, probe = [], []
    for row in range(N):
        a = int(rng.integers(4, 25)) * stride
        b = int(rng.integers(4, 25)) * stride
        table.append((row, 0, a))
        table.append((row, a + 2 * stride, b))
    # shuffle across rows (per-row order must stay ascending)
    order = rng.permutation(N)
    table = [t for r in order for t in (table[2 * r], table[2 * r + 1])]
    eng = capi.Engine(cfg, ws)
    got = eng.call_var(x, table)
    for k in list(rng.integers(0, len(table), size=24)) + [i for i, t in enumerate(table) if t[0] in (4095, 4096)]:
        row, s0, ln = table[k]
        xa = np.zeros((256, t_in), np.float16)
        xa[0, :ln] = x[row, s0:s0 + ln]
        alone = eng.call_var(xa, [(0, 0, ln)])[0]
        assert got[k][0] == alone[0] and got[k][1] == alone[1] and (got[k][2] == alone[2]).all(), f"chunk {k} (row {row})"
    eng.close()


def test_bench_single_process_hip_all_leg_runs():
    """bench.py's extra.single_process_hip_all (reported at --gpus N > 1 by rank 0): the same function on whatever devices are
    visible (one here) — the call path of the 8-GPU line must not be exercised for the first time by the driver's SCALE run."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = _cfg(128, 4)
    cfg.chunk_size, cfg.overlap = 1200, 120
    cfg.normalise_basecaller_params()
    ws = synth.make_weights(cfg, seed=3)
    sp = bench.single_process_hip_all(cfg, ws, 64, cfg.chunk_size, capi.device_count(), 1.0e6, nb=4)
    assert sp["devices"] == capi.device_count() and sp["samples_per_s"] > 0 and sp["bases"] > 0
