"""The reference-side binding (integration/HipModelRunnerAdapter.h) as COMPILED CODE: oracle/Makefile.ref builds
integration/adapter_test.cpp against the reference's real basecall/ModelRunnerBase.h, config/BasecallModelConfig.h,
DecodedChunk.h, utils/stats.h and libtorch (oracle/_ref/libmibc_adapter_test.so; built in this container, travels
to the GPU box).  CPU: the library loads and exports the driver.  GPU: runners created by create_hip_basecall_runners
are driven through dorado::basecall::ModelRunnerBase::accept_chunk(int, const at::Tensor&) / call_chunks and must
return exactly what the C-ABI returns for the same chunks; runner order [devices][runners][chunk_sizes]."""
import ctypes as C
import os

import numpy as np
import pytest

from dorado_amd import capi, config, hostapi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libmibc_adapter_test.so")
needs_so = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libmibc_adapter_test.so not built "
                                                              "(needs /root/reference: make -C oracle -f Makefile.ref)")


def _load():
    import torch  # noqa: F401  libtorch first
    capi.lib()
    C.CDLL(hostapi.LIB_PATH, mode=C.RTLD_GLOBAL)
    L = C.CDLL(SO)
    L.adapter_last_error.restype = C.c_char_p
    return L


@needs_so
def test_adapter_library_loads_and_exports():
    L = _load()
    assert hasattr(L, "adapter_run") and hasattr(L, "adapter_last_error")


@needs_so
@pytest.mark.gpu
@pytest.mark.parametrize("pipeline,model", [(1, "lstm"), (0, "lstm"), (1, "tx")])
def test_adapter_through_reference_interface(pipeline, model):
    """pipeline: 0 simplex_low_latency, 1 simplex (ModelRunnerBase.h:41)."""
    L = _load()
    if model == "lstm":
        cfg = config.tiny(128, 4)
        cfg.lstm_layers = 5
        cfg.chunk_size, cfg.overlap = 1200, 120
        cfg.qscale, cfg.qbias = 1.05, -0.3
    else:
        cfg = config.tiny_tx()
    cfg.normalise_basecaller_params()
    ws = [np.ascontiguousarray(w, np.float32) for w in synth.make_weights(cfg, seed=91)]
    n, t_in = 24, cfg.chunk_size
    x = synth.make_signal(n, t_in, seed=92)
    eng = capi.Engine(cfg, ws)
    T = eng.output_steps(t_in)
    batch = 64 if model == "lstm" else 32
    xb = np.zeros((batch, t_in), np.float16)
    xb[:n] = x
    want = eng.call(xb)[:n]
    eng.close()

    d = cfg.to_desc()
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    numel = (C.c_int64 * len(ws))(*[w.size for w in ws])
    seq, qs = np.zeros((n, T), np.uint8), np.zeros((n, T), np.uint8)
    mv = np.zeros((n, T), np.uint8)
    info = (C.c_int * 16)()
    name = C.create_string_buffer(128)
    xc = np.ascontiguousarray(x)
    rc = L.adapter_run(C.byref(d), arr, numel, len(ws), b"hip:0", pipeline, 2, t_in, cfg.overlap, batch,
                       C.c_float(cfg.qscale), C.c_float(cfg.qbias), xc.ctypes.data_as(C.c_void_p), n, T,
                       seq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p),
                       info, name, 128)
    assert rc == 0, L.adapter_last_error().decode()
    sizes = hostapi.simplex_chunk_sizes(cfg, t_in, cfg.overlap) if pipeline == 1 else [t_in]
    # [devices][runners][chunk_sizes]: 1 device x 2 runners x len(sizes), chunk sizes repeating in runner order
    assert info[0] == 2 * len(sizes) and info[1] == 1
    assert [info[2 + i] for i in range(min(4, info[0]))] == (sizes * 2)[:min(4, info[0])]
    assert info[6] == 0 and info[7] == (1 if pipeline == 0 else 0) and info[8] == batch
    assert info[11] == 1, "config() must stay valid after the creation parameters are gone (runner-owned copy)"
    assert (info[9], info[10]) == ((350, 350) if pipeline == 0 else (300000, 30000))
    assert name.value.decode().startswith("HipModelRunner_")
    for i, (s, q, m) in enumerate(want):
        assert seq[i, :len(s)].tobytes().decode() == s and not seq[i, len(s):].any()
        assert qs[i, :len(q)].tobytes().decode() == q
        assert (mv[i] == m).all()


@needs_so
@pytest.mark.gpu
@pytest.mark.parametrize("lstm_mode", ["CUTLASS_TNC_F16", None])
def test_adapter_variable_chunks_through_reference_interface(lstm_mode, monkeypatch):
    """lstm_mode: the reference's DORADO_LSTM_MODE override (nn/ConvStack.cpp:76-88), which the adapter honours; None = the
    reference's rule (round 6: this tanh-conv lstm_size-256 model then runs its LSTM stack in int8, nn/ConvStack.cpp:69-73 —
    variable chunk sizes over the int8 LSTM is the reference's default GPU mode).
    BasecallerCreationParams::variable_chunk_sizes = true: variable_chunk_sizes() is reported, chunks of any
    stride-multiple length go in through accept_chunk (index ignored, CudaModelRunner.cpp:21-32) and come back from one
    call_chunks in order; every chunk must equal the engine's call of that chunk ALONE (zero state, zero padding)."""
    L = _load()
    cfg = config.tiny(256, 4)
    cfg.lstm_layers = 3
    cfg.chunk_size, cfg.overlap = 1200, 120
    cfg.normalise_basecaller_params()
    ws = [np.ascontiguousarray(w, np.float32) for w in synth.make_weights(cfg, seed=71)]
    if lstm_mode is None:
        monkeypatch.delenv("DORADO_LSTM_MODE", raising=False)
        assert cfg.reference_gpu_lstm_int8()
        cfg.lstm_quant = True               # what the adapter's descriptor will say; the stand-alone calls below use the same
    else:
        monkeypatch.setenv("DORADO_LSTM_MODE", lstm_mode)
    stride, t_in, batch = cfg.stride, cfg.chunk_size, 64
    rng = np.random.default_rng(5)
    # > batch chunks, many short ones (several per row) and a few just over half a row (one per row: forces the
    # overflow batch of HipModelRunner::call_chunks as well)
    lens = [int(v) * stride for v in rng.integers(2, t_in // stride // 3, size=70)] + [(t_in // stride // 2 + 1) * stride] * 40
    rng.shuffle(lens)
    sig = synth.make_signal(1, int(sum(lens)), seed=72)[0]
    T = t_in // stride
    eng = capi.Engine(cfg, ws)
    want, pos = [], 0
    for ln in lens:                       # the chunk alone, in row 0 of an otherwise empty batch
        xb = np.zeros((batch, t_in), np.float16)
        xb[0, :ln] = sig[pos:pos + ln]
        want.append(eng.call_var(xb, [(0, 0, ln)])[0])
        pos += ln
    eng.close()
    d = cfg.to_desc()
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    numel = (C.c_int64 * len(ws))(*[w.size for w in ws])
    n = len(lens)
    seq, qs, mv = np.zeros((n, T), np.uint8), np.zeros((n, T), np.uint8), np.zeros((n, T), np.uint8)
    mlen = np.zeros(n, np.int64)
    clens = (C.c_int64 * n)(*lens)
    info = (C.c_int * 8)()
    rc = L.adapter_run_variable(C.byref(d), arr, numel, len(ws), b"hip:0", t_in, cfg.overlap, batch, C.c_float(cfg.qscale),
                                C.c_float(cfg.qbias), np.ascontiguousarray(sig).ctypes.data_as(C.c_void_p), clens, n, T,
                                seq.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p),
                                mv.ctypes.data_as(C.c_void_p), mlen.ctypes.data_as(C.c_void_p), info)
    assert rc == 0, L.adapter_last_error().decode()
    # variable mode: batch_size() is the budget offered to BasecallerNode (rows x CallerParams::variable_batch_fill, whole
    # 32-row spans), not the row count of the engine batch
    assert info[0] == 1 and info[1] == 32 and info[2] == t_in
    for i, (s, q, m) in enumerate(want):
        assert mlen[i] == lens[i] // stride == len(m)
        assert (mv[i, :len(m)] == m).all(), f"chunk {i}: moves differ from the stand-alone call"
        assert seq[i, :len(s)].tobytes().decode() == s and qs[i, :len(q)].tobytes().decode() == q


@needs_so
@pytest.mark.gpu
def test_adapter_honours_creation_params():
    """memory_limit_fraction caps the automatic batch size (CudaCaller.cpp:434-439); run_batchsize_benchmarks selects
    the timing sweep with batch_size_time_penalty (:603-627)."""
    L = _load()
    cfg = config.tiny(128, 4)
    cfg.chunk_size, cfg.overlap = 2400, 120
    cfg.normalise_basecaller_params()
    ws = [np.ascontiguousarray(w, np.float32) for w in synth.make_weights(cfg, seed=3)]
    d = cfg.to_desc()
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    numel = (C.c_int64 * len(ws))(*[w.size for w in ws])
    free_b, _ = capi.device_memory(0)

    def auto(frac, penalty=0.05, bench=0):
        out = C.c_int(0)
        rc = L.adapter_auto_batch(C.byref(d), arr, numel, len(ws), b"hip:0", cfg.chunk_size, cfg.overlap, C.c_float(frac),
                                  C.c_float(penalty), bench, C.byref(out))
        return rc, out.value

    rc, full = auto(0.8)
    assert rc == 0 and full == 256 * 64, L.adapter_last_error().decode()       # the knee: one workgroup per CU
    # a fraction that leaves (after the 1 GB reserve) room for only a few hundred chunks
    eng = capi.Engine(cfg, ws)
    per, fixed = eng.query_memory(cfg.chunk_size)
    eng.close()
    frac = ((1 << 30) + fixed + 1000 * per) / free_b
    rc, small = auto(frac)
    assert rc == 0 and 64 <= small <= 1000 and small % 64 == 0, (small, L.adapter_last_error().decode())
    # less than one batch granule inside the limit: warn and fall back to the smallest batch, as the reference falls back
    # to its default batch (CudaCaller.cpp:441-445), instead of refusing to start
    rc, tiny = auto(((1 << 30) * 0.5) / free_b)
    assert rc == 0 and tiny == 64
    rc, swept = auto(0.8, penalty=0.5, bench=1)      # a generous penalty accepts a smaller batch than the best one
    assert rc == 0 and 64 <= swept <= 2 * full and swept % 64 == 0


@needs_so
@pytest.mark.gpu
@pytest.mark.parametrize("variable,lstm_mode", [(0, None), (1, "CUTLASS_TNC_F16"), (1, None)])
def test_reference_basecaller_node_drives_the_engine(variable, lstm_mode, monkeypatch):
    """The drop-in claim end to end: the reference's OWN BasecallerNode (read_pipeline/nodes/BasecallerNode.cpp compiled in place
    with its MessageSink / chunk / stitch sources — integration/basecaller_node_test.cpp) chunks, batches, times out, stitches;
    its runners are the HipModelRunnerAdapter objects create_hip_basecall_runners returns ([device][runner][chunk size], two
    chunk-size queues for the simplex pipeline).  Reads go in through push_message, called reads come out of a sink.  Every read
    must equal what this repo's own node (host::SimplexBasecaller, the mirror of that file) returns for it: same chunk plan,
    same stitching, same bytes."""
    L = _load()
    # variable chunk sizes: the adapter applies the reference's model rule (api/runner_creation.cpp:24-44): lstm_size > 128,
    # else the runners report fixed chunks and the node repeat-pads short reads (which the first version of this test ran into)
    cfg = config.tiny(256, 4) if variable else config.tiny(128, 4)
    cfg.lstm_layers = 3 if variable else 5
    cfg.chunk_size, cfg.overlap = 1200, 120
    cfg.qscale, cfg.qbias = 1.05, -0.3
    cfg.normalise_basecaller_params()
    ws = [np.ascontiguousarray(w, np.float32) for w in synth.make_weights(cfg, seed=17)]
    # the adapter picks the LSTM arithmetic by the reference's rule (nn/ConvStack.cpp:60-89; DORADO_LSTM_MODE overrides it):
    # the lstm_size-256 model runs int8 unless the override says f16; this repo's own node gets the same descriptor
    if lstm_mode is None:
        monkeypatch.delenv("DORADO_LSTM_MODE", raising=False)
        cfg.lstm_quant = cfg.reference_gpu_lstm_int8()
        assert cfg.lstm_quant == bool(variable)
    else:
        monkeypatch.setenv("DORADO_LSTM_MODE", lstm_mode)
    lens = [300, 594, 600, 1200, 1206, 2500, 3343, 5010, 809, 4106, 7777, 12000, 312 if variable else 66]
    reads = [synth.make_signal(1, L_, seed=300 + i)[0] for i, L_ in enumerate(lens)]
    d = cfg.to_desc()
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    numel = (C.c_int64 * len(ws))(*[w.size for w in ws])
    n = len(reads)
    pitch = max(lens) // cfg.stride + 8
    seq, qs, mv = (np.zeros((n, pitch), np.uint8) for _ in range(3))
    sl, ml = np.zeros(n, np.int64), np.zeros(n, np.int64)
    st = (C.c_double * 8)()
    sig = np.ascontiguousarray(np.concatenate(reads).astype(np.float16))
    rl = np.array(lens, np.int64)
    RESTART_AFTER = 0       # (terminate + restart of the node and its runners: the CPU test over the engine double covers it)
    rc = L.adapter_run_basecaller_node(C.byref(d), arr, numel, len(ws), b"hip:0", 2, cfg.chunk_size, cfg.overlap, 64, variable,
                                       C.c_float(cfg.qscale), C.c_float(cfg.qbias), sig.ctypes.data_as(C.c_void_p),
                                       rl.ctypes.data_as(C.c_void_p), n, pitch, seq.ctypes.data_as(C.c_void_p),
                                       qs.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p),
                                       sl.ctypes.data_as(C.c_void_p), ml.ctypes.data_as(C.c_void_p), st, RESTART_AFTER)
    assert rc == 0, L.adapter_last_error().decode()
    assert st[4] == variable                           # the runners really are in the mode under test
    if variable:
        want, _ = hostapi.basecall_reads(cfg, ws, reads, device="hip:0", num_runners=2, batch_size=64, variable_chunks=True)
    else:
        want, _ = hostapi.basecall_reads(cfg, ws, reads, device="hip:0", num_runners=2, batch_size=64, two_queues=True)
    bases = 0
    for r in range(n):
        got_seq = seq[r, :sl[r]].tobytes().decode()
        got_qs = qs[r, :sl[r]].tobytes().decode()
        assert ml[r] == lens[r] // cfg.stride, (r, lens[r])
        assert got_seq == want[r][0] and got_qs == want[r][1], (r, lens[r])
        assert (mv[r, :ml[r]] == want[r][2]).all(), (r, lens[r])
        assert int(mv[r, :ml[r]].sum()) == len(got_seq)
        bases += len(got_seq)
    assert bases > 1000
    assert st[2] == sum(lens)                          # samples_processed as BasecallerNode counts them
    assert st[0] + st[1] >= 1


@needs_so
def test_host_node_equals_reference_node_with_identical_runners():
    """No GPU: the reference's OWN BasecallerNode and this repo's node (host::SimplexBasecaller) get runners that call a chunk by
    the same pure function of its samples (integration/node_cpu_test.cpp), and the same reads — 60 random configurations (stride,
    chunk size, overlap, one or two chunk-size queues, batch size 1 .. 40, 1 .. 3 runners per size) x up to 200 reads of 1 sample
    to 6 chunks, plus the boundary lengths.  Chunk plan, queue choice, repeat-padding, partial batches and stitching must agree:
    every read byte-identical, and both sides must have called the same number of chunks."""
    L = _load()
    rng = np.random.default_rng(7)
    reads_total = bases_total = 0
    for it in range(60):
        stride = int(rng.choice([5, 6, 12, 2]))
        cs = int(rng.integers(40, 700)) * stride
        overlap = int(rng.integers(0, max(1, cs // stride // 3))) * stride
        sizes = [cs]
        if rng.integers(0, 2):
            half = max(stride * (overlap // stride + 1), (cs // 2) // stride * stride)
            if half < cs:
                sizes.append(half)
        batch, runners = int(rng.integers(1, 40)), int(rng.integers(1, 4))
        lens = np.concatenate([rng.integers(1, 6 * cs, int(rng.integers(1, 200))), rng.integers(1, 3 * stride, 5),
                               [cs, cs - 1, cs + 1, sizes[-1], sizes[-1] + 1, 2 * cs - overlap]]).astype(np.int64)
        sig = rng.integers(0, 65535, int(lens.sum())).astype(np.uint16)
        arr = (C.c_int * len(sizes))(*sizes)
        out = (C.c_long * 6)()
        rc = L.node_cpu_compare(sig.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), len(lens), arr, len(sizes),
                                overlap, stride, batch, runners, 5, out)
        cfg = (stride, sizes, overlap, batch, runners, len(lens))
        assert rc == 0, (cfg, L.adapter_last_error().decode())
        assert out[0] == 0, (cfg, "first differing read", out[1], int(lens[out[1]]))
        assert out[3] == out[4] > 0, cfg                 # chunks called by the two nodes' runners
        reads_total += len(lens)
        bases_total += out[2]
    assert reads_total > 4000 and bases_total > 1e6


FAKE_SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libmibc_adapter_fake.so")


@pytest.mark.skipif(not os.path.exists(FAKE_SO), reason="oracle/_ref/libmibc_adapter_fake.so not built (needs /root/reference)")
def test_reference_node_over_the_whole_host_layer_without_a_gpu():
    """No GPU: the reference's OWN BasecallerNode -> HipModelRunnerAdapter -> HipModelRunner -> HipCaller (GPU thread, device
    FIFO, two asynchronous slots) -> C-ABI test double (tools/fake_mibc.cpp), 1500 reads from 6 samples to 12 chunks, fixed (two
    chunk-size queues) and variable chunk sizes (several chunks per batch row, the node's 32-row-span budget against the
    runner's batch_size(), overflow batches), the node and its runners terminated and restarted half way (NodeSmokeTest.cpp's
    restart case) — every read must equal what this repo's node returns over the same double.
    Own process: the double never meets the real libmibc.so (tools/ref_node_over_fake_engine.py)."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "ref_node_over_fake_engine.py"), "1500"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for mode, is_var in (("fixed", 0), ("variable", 1)):
        m = res[mode]
        assert m["differing_reads"] == 0, (mode, m)
        assert m["runners_variable"] == is_var and m["reads"] >= 1500 and m["bases"] > 2e5, (mode, m)
    assert res["variable"]["var_engine_batches"] > 0 and res["variable"]["var_overflow_batches"] > 0
