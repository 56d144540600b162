"""ctypes binding of the C++ host layer (dorado_amd/libmibc_host.so = dorado_amd/host/):
chunking / stitching / device-string parsing (CPU-only entry points) and whole-read basecalling
through create_basecall_runners + SimplexBasecaller (needs a GPU)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import capi
from .config import ModelConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmibc_host.so")
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        capi.lib()  # libmibc.so first (and torch's HIP runtime when available)
        if not os.path.exists(LIB_PATH):
            raise capi.MibcError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.mibch_last_error.restype = C.c_char_p
        L.mibch_generate_chunks.restype = C.c_long
        L.mibch_stitch_chunks.restype = C.c_long
        _lib = L
    return _lib


def generate_chunks(num_samples, chunk_size, stride, overlap):
    cap = 1 << 16
    out = (C.c_uint64 * cap)()
    n = lib().mibch_generate_chunks(C.c_uint64(num_samples), C.c_uint64(chunk_size), C.c_uint64(stride),
                                    C.c_uint64(overlap), out, C.c_long(cap))
    if n < 0:
        raise ValueError(lib().mibch_last_error().decode())
    return [int(out[i]) for i in range(n)]


def parse_device_ids(s: str, num_devices: int):
    ids = (C.c_int * 64)()
    n = C.c_int(0)
    ok = lib().mibch_parse_device_ids(s.encode(), C.c_uint64(num_devices), ids, 64, C.byref(n))
    return bool(ok), [int(ids[i]) for i in range(n.value)]


def stitch_chunks(offsets, raw_chunk_sizes, moves_list, seqs, qstrs, raw_samples, stride):
    n = len(offsets)
    moves = np.concatenate([np.asarray(m, np.uint8) for m in moves_list])
    mlen = np.array([len(m) for m in moves_list], np.int64)
    moff = np.concatenate([[0], np.cumsum(mlen)[:-1]]).astype(np.int64)
    slen = np.array([len(s) for s in seqs], np.int64)
    soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.int64)
    cap = int(slen.sum()) + 8
    so, qo = C.create_string_buffer(cap), C.create_string_buffer(cap)
    mo = np.zeros(int(mlen.sum()) + 8, np.uint8)
    nm = C.c_int64(0)
    io, rc = np.asarray(offsets, np.int64), np.asarray(raw_chunk_sizes, np.int64)
    L = lib().mibch_stitch_chunks(C.c_int(n), io.ctypes.data_as(_i64p), rc.ctypes.data_as(_i64p),
                                  moves.ctypes.data_as(_u8p), moff.ctypes.data_as(_i64p),
                                  mlen.ctypes.data_as(_i64p), "".join(seqs).encode(),
                                  "".join(qstrs).encode(), soff.ctypes.data_as(_i64p),
                                  slen.ctypes.data_as(_i64p), C.c_int64(raw_samples), C.c_int(stride),
                                  so, qo, mo.ctypes.data_as(_u8p), C.byref(nm))
    if L < 0:
        raise ValueError(lib().mibch_last_error().decode())
    return so.raw[:L].decode(), qo.raw[:L].decode(), mo[: nm.value].copy()


def generate_variable_chunks(num_samples, chunk_size, stride, overlap):
    cap = 1 << 17
    out = (C.c_uint64 * (2 * cap))()
    lib().mibch_generate_variable_chunks.restype = C.c_long
    n = lib().mibch_generate_variable_chunks(C.c_uint64(num_samples), C.c_uint64(chunk_size), C.c_uint64(stride),
                                             C.c_uint64(overlap), out, C.c_long(cap))
    if n < 0:
        raise ValueError(lib().mibch_last_error().decode())
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def simplex_chunk_sizes(cfg: ModelConfig, requested_chunk_size, overlap):
    """CudaCaller.cpp:382-413: {chunk, 0.5 x chunk} normalised to the chunk granularity, largest first."""
    out = (C.c_int * 8)()
    d = cfg.to_desc()
    n = lib().mibch_simplex_chunk_sizes(C.byref(d), int(requested_chunk_size), int(overlap), out, 8)
    return [int(out[i]) for i in range(n)]


def get_chunk_queue_idx(chunk_sizes, read_raw_size):
    """BasecallerNode::get_chunk_queue_idx (BasecallerNode.cpp:81-94)."""
    arr = (C.c_uint64 * len(chunk_sizes))(*[int(v) for v in chunk_sizes])
    return int(lib().mibch_get_chunk_queue_idx(arr, len(chunk_sizes), C.c_uint64(int(read_raw_size))))


def auto_batch_size(cfg: ModelConfig, weights, mode=0, device=0):
    """Batch size HipCaller chooses for cfg.chunk_size: mode 0 = known knee (one LSTM workgroup / cluster slot per CU,
    bounded by memory), mode -1 = the reference's timing sweep.  Returns (batch, [(batch, ms_per_chunk), ...])."""
    d = cfg.to_desc()
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    opts = capi.DecodeOptsC(32, 100.0, 2.0, cfg.qbias, cfg.qscale)
    chosen, nt = C.c_int(), C.c_int()
    tim = (C.c_double * 32)()
    rc = lib().mibch_auto_batch_size(C.byref(d), arr, len(ws), device, cfg.chunk_size, mode, C.byref(opts),
                                     C.byref(chosen), tim, 16, C.byref(nt))
    if rc != 0:
        raise capi.MibcError(lib().mibch_last_error().decode())
    return chosen.value, [(int(tim[2 * i]), float(tim[2 * i + 1])) for i in range(min(nt.value, 16))]


def model_stride(cfg: ModelConfig):
    d = cfg.to_desc()
    return int(lib().mibch_model_stride(C.byref(d)))


def basecall_reads(cfg: ModelConfig, weights, reads_f16, device="hip:0", num_runners=2, batch_size=64,
                   beam_width=32, variable_chunks=False, two_queues=False):
    """reads_f16: list of 1-D f16 arrays.  Returns (list of (seq, qstr, moves, chunk_offsets), stats).
    two_queues: the reference's extra 0.5x chunk-size queue ([device][runner][chunk_size] runners)."""
    L = lib()
    d = cfg.to_desc()
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    opts = capi.DecodeOptsC(beam_width, 100.0, 2.0, cfg.qbias, cfg.qscale)
    sig = np.ascontiguousarray(np.concatenate(reads_f16).astype(np.float16))
    lens = np.array([len(r) for r in reads_f16], np.int64)
    n = len(reads_f16)
    tot_steps = int(sum(l // cfg.stride + 2 for l in lens))
    seq = C.create_string_buffer(tot_steps + 8)
    qs = C.create_string_buffer(tot_steps + 8)
    mv = np.zeros(tot_steps + 8, np.uint8)
    sl = np.zeros(n, np.int64)
    ml = np.zeros(n, np.int64)
    max_off = int(sum(l // max(1, cfg.chunk_size // 2 - cfg.overlap) + 3 for l in lens))
    offs = np.zeros(max_off, np.int64)
    noff = np.zeros(n, np.int64)
    stats = (C.c_double * 4)()
    fn = L.mibch_basecall_reads_variable if variable_chunks else L.mibch_basecall_reads
    args = [C.byref(d), arr, len(ws), device.encode(), num_runners, cfg.chunk_size,
            cfg.overlap, batch_size, C.byref(opts), sig.ctypes.data_as(C.c_void_p),
            lens.ctypes.data_as(_i64p), n, seq, qs, sl.ctypes.data_as(_i64p),
            mv.ctypes.data_as(_u8p), ml.ctypes.data_as(_i64p),
            offs.ctypes.data_as(_i64p), noff.ctypes.data_as(_i64p), stats]
    if two_queues:
        fn = L.mibch_basecall_reads_two_queues
        args.append((C.c_int * 4)())
    rc = fn(*args)
    if rc != 0:
        raise capi.MibcError(L.mibch_last_error().decode())
    out = []
    so = mo = oo = 0
    for r in range(n):
        out.append((seq.raw[so:so + sl[r]].decode(), qs.raw[so:so + sl[r]].decode(),
                    mv[mo:mo + ml[r]].copy(), offs[oo:oo + noff[r]].tolist()))
        so += int(sl[r]); mo += int(ml[r]); oo += int(noff[r])
    return out, {"samples_processed": stats[0], "samples_incl_padding": stats[1],
                 "batches_called": stats[2], "partial_batches_called": stats[3]}


def bench_through_host(cfg: ModelConfig, weights, reads_f16, n_warm, n_reads, device="hip:0", num_runners=2,
                       batch_size=0, beam_width=32, two_queues=False):
    """Throughput of the C++ host path (SimplexBasecaller, `num_runners` runners per device and batch dimension, two
    batches in flight per device) on synthetic reads: reads_f16 [n_distinct, read_len] f16 are cycled n_reads times;
    device may name several GPUs ("hip:all": one process, one HipCaller per device).  Returns dict(samples_per_s = read
    samples per second (overlap counted once, as BasecallerNode's samples_processed), samples_incl_padding_per_s,
    seconds, batches, partial_batches, bases, devices)."""
    L = lib()
    d = cfg.to_desc()
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    opts = capi.DecodeOptsC(beam_width, 100.0, 2.0, cfg.qbias, cfg.qscale)
    sig = np.ascontiguousarray(reads_f16, np.float16)
    out = (C.c_double * 8)()
    rc = L.mibch_bench_through_host(C.byref(d), arr, len(ws), device.encode(), num_runners, cfg.chunk_size, cfg.overlap,
                                    batch_size, C.byref(opts), sig.ctypes.data_as(C.c_void_p), int(sig.shape[0]),
                                    C.c_int64(sig.shape[1]), C.c_int64(n_warm), C.c_int64(n_reads),
                                    C.c_int(1 if two_queues else 0), out)
    if rc != 0:
        raise capi.MibcError(L.mibch_last_error().decode())
    return {"samples_per_s": out[0], "seconds": out[1], "batches": out[2], "bases": out[3],
            "samples_incl_padding_per_s": out[4], "devices": int(out[5]), "partial_batches": out[6]}


def bench_through_host_variable(cfg: ModelConfig, weights, signals_f16, read_lens, n_warm, device="hip:0", num_runners=2,
                                batch_size=0, beam_width=32, variable=True):
    """Throughput of the variable-chunk host path (SimplexBasecaller::basecall_variable -> mibc_call_var_async): read r is
    the first read_lens[r] samples of signals_f16[r % n_distinct]; the first n_warm reads are untimed.  variable=False: the
    SAME read set through the fixed-chunk path (SimplexBasecaller::basecall).  Returns
    dict(samples_per_s = read samples per second, seconds, batches, bases, samples_incl_padding_per_s, devices)."""
    L = lib()
    d = cfg.to_desc()
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    opts = capi.DecodeOptsC(beam_width, 100.0, 2.0, cfg.qbias, cfg.qscale)
    sig = np.ascontiguousarray(signals_f16, np.float16)
    lens = np.ascontiguousarray(np.minimum(np.asarray(read_lens, np.int64), sig.shape[1]))
    out = (C.c_double * 8)()
    rc = L.mibch_bench_through_host_mixed(C.byref(d), arr, len(ws), device.encode(), num_runners, cfg.chunk_size, cfg.overlap,
                                          batch_size, C.byref(opts), sig.ctypes.data_as(C.c_void_p), int(sig.shape[0]),
                                          C.c_int64(sig.shape[1]), lens.ctypes.data_as(C.c_void_p), C.c_int64(n_warm),
                                          C.c_int64(len(lens) - n_warm), C.c_int(int(bool(variable))), out)
    if rc != 0:
        raise capi.MibcError(L.mibch_last_error().decode())
    return {"samples_per_s": out[0], "seconds": out[1], "batches": out[2], "bases": out[3],
            "samples_incl_padding_per_s": out[4], "devices": int(out[5])}


# ---------------------------------------------------------------- ScalerNode host half
def pa_read_scaling(standardise, mean, stdev, scaling, offset, open_pore_level=float("nan"),
                    flow_cell_product_code=""):
    """ScalerNode.cpp:186-227 (strategy PA) -> dict(shift, scale, open_pore_adjustment, scale_pa, shift_pa)."""
    out = (C.c_float * 5)()
    rc = lib().mibch_pa_read_scaling(C.c_int(int(standardise)), C.c_float(mean), C.c_float(stdev),
                                     C.c_float(scaling), C.c_float(offset), C.c_float(open_pore_level),
                                     flow_cell_product_code.encode(), out)
    if rc != 0:
        raise ValueError(lib().mibch_last_error().decode())
    return dict(zip(("shift", "scale", "open_pore_adjustment", "scale_pa", "shift_pa"), [float(v) for v in out]))


def trim_signal(scaled_f16, threshold=2.4, window_size=40, min_elements=3):
    s = np.ascontiguousarray(scaled_f16, np.float16)
    return int(lib().mibch_trim_signal(s.ctypes.data_as(C.c_void_p), C.c_int(s.size), C.c_float(threshold),
                                       C.c_int(window_size), C.c_int(min_elements)))


def dna_trim_start(standardise, scaled_f16):
    s = np.ascontiguousarray(scaled_f16, np.float16)
    return int(lib().mibch_dna_trim_start(C.c_int(int(standardise)), s.ctypes.data_as(C.c_void_p),
                                          C.c_uint64(s.size)))


def rna_adapter_pos(raw_i16):
    """ScalerNode.cpp:58-107 (determine_rna_adapter_pos) on a raw int16 read: where the DNA adapter of a dRNA read ends, 0 if
    no median jump is found."""
    s = np.ascontiguousarray(raw_i16, np.int16)
    return int(lib().mibch_rna_adapter_pos(s.ctypes.data_as(C.c_void_p), C.c_int(s.size)))


def rna_trim(raw_i16, has_rna_based_adapters=False):
    """ScalerNode.cpp:157-184 -> dict(trim_start, rna_adapter_end_signal_pos)."""
    s = np.ascontiguousarray(raw_i16, np.int16)
    out = (C.c_int * 2)()
    lib().mibch_rna_trim(s.ctypes.data_as(C.c_void_p), C.c_uint64(s.size), C.c_int(int(has_rna_based_adapters)), out)
    return {"trim_start": int(out[0]), "rna_adapter_end_signal_pos": int(out[1])}


SCALING_STRATEGIES = {"med_mad": 0, "quantile": 1, "pa": 2}   # config::ScalingStrategy order


def scaler_kwargs(cfg: ModelConfig) -> dict:
    """What ScalerNode is constructed with for a model (ScalerNode(config.signal_norm_params, config.sample_type, ...),
    e.g. cli/basecaller.cpp), as keyword arguments of scaler_node / scaler_node_ops: from config.load_model_config's
    signal_norm and sample_type."""
    sn = cfg.signal_norm
    return {"strategy": sn.strategy, "quantile": (sn.quantile_a, sn.quantile_b, sn.shift_multiplier, sn.scale_multiplier),
            "standardisation": (sn.standardise, sn.mean, sn.stdev),
            # ScalerNode.cpp:157: m_is_rna_model = (model_type == SampleType::RNA004) — RNA002 reads get no adapter cut and go
            # through the DNA trim heuristic
            "is_rna_model": cfg.sample_type == "RNA004"}


def scaler_node(cfg: ModelConfig, weights, raw_i16, strategy="quantile", quantile=(0.2, 0.9, 0.51, 0.53),
                standardisation=(False, 0.0, 1.0), is_rna_model=False, has_rna_based_adapters=False,
                scaling=1.0, offset=0.0, open_pore_level=float("nan"), flow_cell_product_code="", device="hip:0",
                want_signal=True):
    """One raw int16 read through the host mirror of ScalerNode::input_thread_fn (ScalerNode.cpp:144-267; statistics and the
    sample map on the device) -> dict(signal (np.float16, scaled + trimmed; None without want_signal), n_out, shift, scale,
    open_pore_adjustment, scale_pa, shift_pa, num_trimmed_samples, rna_adapter_end_signal_pos, first_sample)."""
    L = lib()
    d = cfg.to_desc()
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    x = np.ascontiguousarray(raw_i16, np.int16)
    p7 = (C.c_float * 7)(*quantile, float(bool(standardisation[0])), standardisation[1], standardisation[2])
    cal = (C.c_float * 3)(scaling, offset, open_pore_level)
    out = np.empty(max(x.size, 1), np.uint16) if want_signal else None
    on = C.c_uint64()
    f5 = (C.c_float * 5)()
    i3 = (C.c_int * 3)()
    rc = L.mibch_scaler_node(C.byref(d), arr, len(ws), device.encode(), C.c_int(SCALING_STRATEGIES[strategy]), p7,
                             C.c_int(int(is_rna_model)), C.c_int(int(has_rna_based_adapters)),
                             x.ctypes.data_as(C.c_void_p), C.c_uint64(x.size), cal, flow_cell_product_code.encode(),
                             out.ctypes.data_as(C.c_void_p) if want_signal else None, C.byref(on), f5, i3)
    if rc != 0:
        raise capi.MibcError(L.mibch_last_error().decode())
    return {"signal": out[:on.value].view(np.float16).copy() if want_signal else None, "n_out": int(on.value),
            "shift": float(f5[0]), "scale": float(f5[1]), "open_pore_adjustment": float(f5[2]),
            "scale_pa": float(f5[3]), "shift_pa": float(f5[4]), "num_trimmed_samples": int(i3[0]),
            "rna_adapter_end_signal_pos": int(i3[1]), "first_sample": int(i3[2])}


_STATS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))
_SCALE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_void_p)


def scaler_node_ops(stats, scale, raw_i16, strategy="quantile", quantile=(0.2, 0.9, 0.51, 0.53),
                    standardisation=(False, 0.0, 1.0), is_rna_model=False, has_rna_based_adapters=False, scaling=1.0,
                    offset=0.0, open_pore_level=float("nan"), flow_cell_product_code="", want_signal=True):
    """scaler_node with the caller's own two passes over the samples (host::ScalerOps; no device): stats(x int16, strategy
    name, params4) -> (shift, scale); scale(x int16, shift, scale) -> np.float16.  Same dict as scaler_node."""
    L = lib()
    x = np.ascontiguousarray(raw_i16, np.int16)

    def _stats(ptr, n, strat, p4, out2):
        v = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), (int(n),)) if n else np.zeros(0, np.int16)
        sh, sc = stats(v, "med_mad" if strat == 0 else "quantile", tuple(p4[i] for i in range(4)))
        out2[0], out2[1] = sh, sc

    def _scale(ptr, n, shift, sc, outp):
        v = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), (int(n),)) if n else np.zeros(0, np.int16)
        y = np.ascontiguousarray(scale(v, shift, sc), np.float16)
        C.memmove(outp, y.ctypes.data, int(n) * 2)

    p7 = (C.c_float * 7)(*quantile, float(bool(standardisation[0])), standardisation[1], standardisation[2])
    cal = (C.c_float * 3)(scaling, offset, open_pore_level)
    out = np.empty(max(x.size, 1), np.uint16) if want_signal else None
    on = C.c_uint64()
    f5 = (C.c_float * 5)()
    i3 = (C.c_int * 3)()
    rc = L.mibch_scaler_node_ops(_STATS_FN(_stats), _SCALE_FN(_scale), C.c_int(SCALING_STRATEGIES[strategy]), p7,
                                 C.c_int(int(is_rna_model)), C.c_int(int(has_rna_based_adapters)),
                                 x.ctypes.data_as(C.c_void_p), C.c_uint64(x.size), cal, flow_cell_product_code.encode(),
                                 out.ctypes.data_as(C.c_void_p) if want_signal else None, C.byref(on), f5, i3)
    if rc != 0:
        raise ValueError(L.mibch_last_error().decode())
    return {"signal": out[:on.value].view(np.float16).copy() if want_signal else None, "n_out": int(on.value),
            "shift": float(f5[0]), "scale": float(f5[1]), "open_pore_adjustment": float(f5[2]),
            "scale_pa": float(f5[3]), "shift_pa": float(f5[4]), "num_trimmed_samples": int(i3[0]),
            "rna_adapter_end_signal_pos": int(i3[1]), "first_sample": int(i3[2])}


def basecall_raw_reads(cfg: ModelConfig, weights, reads_i16, shift_scale, trim_start=None, device="hip:0",
                       num_runners=2, batch_size=64, beam_width=32):
    """reads_i16: list of RAW int16 reads; shift_scale [n,2]; trim_start [n] samples cut from the front
    (ScalerNode's num_trimmed_samples).  Scaling runs on the device, fused into conv1."""
    L = lib()
    d = cfg.to_desc()
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
    opts = capi.DecodeOptsC(beam_width, 100.0, 2.0, cfg.qbias, cfg.qscale)
    sig = np.ascontiguousarray(np.concatenate(reads_i16).astype(np.int16))
    lens = np.array([len(r) for r in reads_i16], np.int64)
    n = len(reads_i16)
    ss = np.ascontiguousarray(shift_scale, np.float32).reshape(n, 2)
    ts = np.zeros(n, np.int64) if trim_start is None else np.ascontiguousarray(trim_start, np.int64)
    tot_steps = int(sum(l // cfg.stride + 2 for l in lens))
    seq = C.create_string_buffer(tot_steps + 8)
    qs = C.create_string_buffer(tot_steps + 8)
    mv = np.zeros(tot_steps + 8, np.uint8)
    sl = np.zeros(n, np.int64)
    ml = np.zeros(n, np.int64)
    max_off = int(sum(l // (cfg.chunk_size - cfg.overlap) + 3 for l in lens))
    offs = np.zeros(max_off, np.int64)
    noff = np.zeros(n, np.int64)
    stats = (C.c_double * 4)()
    rc = L.mibch_basecall_raw_reads(C.byref(d), arr, len(ws), device.encode(), num_runners, cfg.chunk_size,
                                    cfg.overlap, batch_size, C.byref(opts), sig.ctypes.data_as(C.c_void_p),
                                    lens.ctypes.data_as(_i64p), ss.ctypes.data_as(C.c_void_p),
                                    ts.ctypes.data_as(_i64p), n, seq, qs, sl.ctypes.data_as(_i64p),
                                    mv.ctypes.data_as(_u8p), ml.ctypes.data_as(_i64p),
                                    offs.ctypes.data_as(_i64p), noff.ctypes.data_as(_i64p), stats)
    if rc != 0:
        raise capi.MibcError(L.mibch_last_error().decode())
    out = []
    so = mo = oo = 0
    for r in range(n):
        out.append((seq.raw[so:so + sl[r]].decode(), qs.raw[so:so + sl[r]].decode(),
                    mv[mo:mo + ml[r]].copy(), offs[oo:oo + noff[r]].tolist()))
        so += int(sl[r]); mo += int(ml[r]); oo += int(noff[r])
    return out, {"samples_processed": stats[0], "samples_incl_padding": stats[1],
                 "batches_called": stats[2], "partial_batches_called": stats[3]}


# ---------------------------------------------------------------- ".tensor" loader
_DTYPES = [np.float16, None, np.float32, np.float64, np.int8, np.uint8, np.int16, np.int32, np.int64, np.bool_]


def load_tensor_file(path, as_float=False):
    """libtorch-free read of a TorchScript ".tensor" archive -> list of (name, ndarray).  bf16 tensors are
    returned as float32 (numpy has no bf16); as_float=True converts everything to float32."""
    L = lib()
    L.mibch_tensor_last_error.restype = C.c_char_p
    n = L.mibch_tensor_open(str(path).encode())
    if n < 0:
        raise ValueError(L.mibch_tensor_last_error().decode())
    out = []
    for i in range(n):
        dt, rank, numel = C.c_int(), C.c_int(), C.c_int64()
        shape = (C.c_int64 * 8)()
        name = C.create_string_buffer(64)
        L.mibch_tensor_info(i, C.byref(dt), C.byref(rank), shape, C.byref(numel), name)
        shp = tuple(shape[d] for d in range(rank.value))
        npdt = _DTYPES[dt.value]
        if as_float or npdt is None:
            a = np.empty(numel.value, np.float32)
            assert L.mibch_tensor_copy_float(i, a.ctypes.data_as(C.c_void_p), C.c_uint64(numel.value)) == 0
        else:
            a = np.empty(numel.value, npdt)
            assert L.mibch_tensor_copy_raw(i, a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes)) == 0
        out.append((name.value.decode(), a.reshape(shp)))
    return out


def model_tensor_names(cfg: ModelConfig):
    """File names in module.parameters() order (basecall/crf_utils.cpp:26-150)."""
    buf = C.create_string_buffer(1 << 16)
    if cfg.tx is not None:
        lib().mibch_model_tensor_names(1, len(cfg.convs), cfg.tx.depth, 0, 0, 0, buf, len(buf))
    else:
        lib().mibch_model_tensor_names(0, len(cfg.convs), cfg.lstm_layers, 0, int(cfg.bias),
                                       int(bool(cfg.out_features)), buf, len(buf))
    return [s for s in buf.value.decode().split("\n") if s]


def load_model_weights(model_dir, cfg: ModelConfig):
    """Weights of a model directory (config.toml + *.tensor) as float32 arrays in the order mibc_create
    expects = the order load_{lstm,tx}_model_weights returns them."""
    import os

    ws = []
    for name in model_tensor_names(cfg):
        ts = load_tensor_file(os.path.join(str(model_dir), name), as_float=True)
        if len(ts) != 1:
            raise ValueError(f"{name}: expected one tensor, found {len(ts)}")
        ws.append(np.ascontiguousarray(ts[0][1]))
    return ws
