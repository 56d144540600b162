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
    assert (info[9], info[10]) == ((350, 350) if pipeline == 0 else (300000, 30000))
    assert name.value.decode().startswith("HipModelRunner_")
    for i, (s, q, m) in enumerate(want):
        assert seq[i, :len(s)].tobytes().decode() == s and not seq[i, len(s):].any()
        assert qs[i, :len(q)].tobytes().decode() == q
        assert (mv[i] == m).all()
