"""ctypes bindings for the CPU oracle (oracle/liboracle.so, the C restatement) and, when
built, the compiled reference (oracle/_ref/libdorado_ref.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
dorado_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libdorado_ref.so")

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


def build(with_ref: bool = True) -> None:
    subprocess.check_call(["make", "-s", "-C", HERE])
    if with_ref and os.path.isdir("/root/reference/dorado"):
        subprocess.check_call(["make", "-s", "-C", HERE, "-f", "Makefile.ref", "-j8"])


def _fp(a):
    return a.ctypes.data_as(_f32p)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(with_ref=False)
        _lib = C.CDLL(ORACLE_SO)
        _lib.orc_beam_search.restype = C.c_float
        _lib.orc_generate_chunks.restype = C.c_long
        _lib.orc_stitch_chunks.restype = C.c_long
        _lib.orc_det_expf.restype = C.c_float
        _lib.orc_det_expf.argtypes = [C.c_float]
        _lib.orc_det_logf.restype = C.c_float
        _lib.orc_det_logf.argtypes = [C.c_float]
    return _lib


_ref = None


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        import torch  # noqa: F401  (libtorch must be loaded first)

        _ref = C.CDLL(REF_SO)
        _ref.ref_last_error.restype = C.c_char_p
    return _ref


# ---------------------------------------------------------------- chunk / stitch
def generate_chunks(num_samples, chunk_size, stride, overlap, use_ref=False):
    cap = 1 << 16
    out = (C.c_uint64 * cap)()
    if use_ref:
        n = ref().ref_generate_chunks(C.c_uint64(num_samples), C.c_uint64(chunk_size),
                                      C.c_uint64(stride), C.c_uint64(overlap), out, cap)
    else:
        n = lib().orc_generate_chunks(C.c_uint64(num_samples), C.c_uint64(chunk_size),
                                      C.c_uint64(stride), C.c_uint64(overlap), out, C.c_long(cap))
    if n < 0:
        raise ValueError("generate_chunks: invalid arguments")
    return [int(out[i]) for i in range(n)]


def stitch_chunks(offsets, raw_chunk_sizes, moves_list, seqs, qstrs, raw_samples, stride):
    n = len(offsets)
    moves = np.concatenate([np.asarray(m, np.uint8) for m in moves_list])
    mlen = np.array([len(m) for m in moves_list], np.int64)
    moff = np.concatenate([[0], np.cumsum(mlen)[:-1]]).astype(np.int64)
    slen = np.array([len(s) for s in seqs], np.int64)
    soff = np.concatenate([[0], np.cumsum(slen)[:-1]]).astype(np.int64)
    seq = "".join(seqs).encode()
    qs = "".join(qstrs).encode()
    cap = int(slen.sum()) + 8
    so = C.create_string_buffer(cap)
    qo = C.create_string_buffer(cap)
    mo = np.zeros(int(mlen.sum()) + 8, np.uint8)
    nm = C.c_int64(0)
    io = np.asarray(offsets, np.int64)
    rc = np.asarray(raw_chunk_sizes, np.int64)
    i64p = C.POINTER(C.c_int64)
    L = lib().orc_stitch_chunks(C.c_int(n), io.ctypes.data_as(i64p), rc.ctypes.data_as(i64p),
                                moves.ctypes.data_as(_u8p), moff.ctypes.data_as(i64p),
                                mlen.ctypes.data_as(i64p), seq, qs, soff.ctypes.data_as(i64p),
                                slen.ctypes.data_as(i64p), C.c_int64(raw_samples), C.c_int(stride),
                                so, qo, mo.ctypes.data_as(_u8p), C.byref(nm))
    return so.raw[:L].decode(), qo.raw[:L].decode(), mo[: nm.value].copy()


# ---------------------------------------------------------------- network
class f16_emulation:
    """with f16_emulation(): the network functions of the C restatement round weights / activations /
    recurrent state to IEEE half exactly where the device path stores them in f16 (oracle.c, g_f16)."""

    def __init__(self, on=True):
        self.on = int(bool(on))

    def __enter__(self):
        self.prev = lib().orc_get_f16_emulation()
        lib().orc_set_f16_emulation(self.on)
        return self

    def __exit__(self, *a):
        lib().orc_set_f16_emulation(self.prev)
        return False


class q8_emulation:
    """with q8_emulation(): the LSTM layers of the C restatement follow the int8 arithmetic of the reference's GPU path
    (nn/LSTMStack.cpp:127-211: per-row quantised weights, round(127 v) activations, integer accumulation); everything else
    rounds as under f16_emulation()."""

    def __init__(self, on=True):
        self.on = int(bool(on))

    def __enter__(self):
        self.prev = lib().orc_get_q8_emulation()
        lib().orc_set_q8_emulation(self.on)
        return self

    def __exit__(self, *a):
        lib().orc_set_q8_emulation(self.prev)
        return False


def _wptrs(weights):
    ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    arr = (_f32p * len(ws))(*[_fp(w) for w in ws])
    return ws, arr


def lstm_crf_forward(cfg, weights, x_f32_NCT, use_ref=False, want_layer=False):
    """x [N, F, T_in] f32 -> scores [N, T, K] f32."""
    x = np.ascontiguousarray(x_f32_NCT, np.float32)
    N, F, T_in = x.shape
    T = T_in // cfg.stride + 2
    K = cfg.outsize
    d = cfg.to_desc()
    ws, arr = _wptrs(weights)
    scores = np.zeros((N, T, K), np.float32)
    if use_ref:
        r = ref()
        T_out = r.ref_forward(C.byref(d), arr, len(ws), _fp(x), N, T_in, _fp(scores),
                              C.c_int64(scores.size))
        if T_out < 0:
            raise RuntimeError(r.ref_last_error().decode())
        return scores.reshape(-1)[: N * T_out * K].reshape(N, T_out, K).copy()
    layer = np.zeros((N, T, cfg.lstm_size), np.float32) if want_layer else None
    T_out = lib().orc_lstm_crf_forward(C.byref(d), arr, _fp(x), N, T_in, _fp(scores),
                                       _fp(layer) if want_layer else None)
    s = scores.reshape(-1)[: N * T_out * K].reshape(N, T_out, K).copy()
    if want_layer:
        return s, layer.reshape(-1)[: N * T_out * cfg.lstm_size].reshape(N, T_out, -1).copy()
    return s


def tx_forward(cfg, weights, x_f32_NCT, use_ref=False, want_tokens=False):
    """Transformer model: x [N, F, T_in] f32 -> scores [N, T_out, K] f32 (+ encoder tokens)."""
    x = np.ascontiguousarray(x_f32_NCT, np.float32)
    N, F, T_in = x.shape
    T_tok = T_in // cfg.conv_stride + 2
    T_out = T_tok * cfg.tx.up_scale_factor
    K = cfg.outsize
    d = cfg.to_desc()
    ws, arr = _wptrs(weights)
    scores = np.zeros((N, T_out, K), np.float32)
    if use_ref:
        r = ref()
        T = r.ref_forward(C.byref(d), arr, len(ws), _fp(x), N, T_in, _fp(scores), C.c_int64(scores.size))
        if T < 0:
            raise RuntimeError(r.ref_last_error().decode())
        return scores.reshape(-1)[: N * T * K].reshape(N, T, K).copy()
    lib().orc_tx_forward.restype = C.c_int
    tok = np.zeros((N, T_tok, cfg.tx.d_model), np.float32) if want_tokens else None
    T = lib().orc_tx_forward(C.byref(d), arr, _fp(x), N, T_in, _fp(scores), _fp(tok) if want_tokens else None)
    if T < 0:
        raise RuntimeError("orc_tx_forward failed")
    s = scores.reshape(-1)[: N * T * K].reshape(N, T, K).copy()
    if want_tokens:
        Tt = T // cfg.tx.up_scale_factor
        return s, tok.reshape(-1)[: N * Tt * cfg.tx.d_model].reshape(N, Tt, -1).copy()
    return s


def forward(cfg, weights, x_f32_NCT, use_ref=False):
    return (tx_forward if cfg.tx is not None else lstm_crf_forward)(cfg, weights, x_f32_NCT, use_ref=use_ref)


# ---------------------------------------------------------------- decoder
def scans(scores_NTK, blank=2.0, use_ref=False, det=0):
    s = np.ascontiguousarray(scores_NTK, np.float32)
    N, T, K = s.shape
    S = K // 4
    fwd = np.zeros((N, T + 1, S), np.float32)
    bwd = np.zeros_like(fwd)
    posts = np.zeros_like(fwd)
    if use_ref:
        r = ref()
        if r.ref_scans(_fp(s), N, T, K, C.c_float(blank), _fp(fwd), _fp(bwd), _fp(posts)) != 0:
            raise RuntimeError(r.ref_last_error().decode())
        return fwd, bwd, posts
    L = lib()
    for n in range(N):
        L.orc_forward_scores(_fp(s[n]), T, S, C.c_float(blank), _fp(fwd[n]), det)
        L.orc_backward_scores(_fp(s[n]), T, S, C.c_float(blank), _fp(bwd[n]), det)
        L.orc_posts(_fp(fwd[n]), _fp(bwd[n]), T, S, _fp(posts[n]), det)
    return fwd, bwd, posts


def decode(scores_NTK, beam_width=32, beam_cut=100.0, blank=2.0, q_shift=0.0, q_scale=1.0,
           use_ref=False, det=0):
    """-> list of (sequence, qstring, moves[T] u8)."""
    s = np.ascontiguousarray(scores_NTK, np.float32)
    N, T, K = s.shape
    moves = np.zeros((N, T), np.uint8)
    seq = np.zeros((N, T), np.uint8)
    qs = np.zeros((N, T), np.uint8)
    lens = np.zeros((N,), np.int32)
    i32p = C.POINTER(C.c_int)
    cp = C.POINTER(C.c_char)
    if use_ref:
        r = ref()
        rc = r.ref_decode(_fp(s), N, T, K, beam_width, C.c_float(beam_cut), C.c_float(blank),
                          C.c_float(q_shift), C.c_float(q_scale), moves.ctypes.data_as(_u8p),
                          seq.ctypes.data_as(cp), qs.ctypes.data_as(cp), lens.ctypes.data_as(i32p))
        if rc != 0:
            raise RuntimeError(r.ref_last_error().decode())
    else:
        lib().orc_decode_batch(_fp(s), N, T, K, beam_width, C.c_float(beam_cut), C.c_float(blank),
                               C.c_float(q_shift), C.c_float(q_scale), moves.ctypes.data_as(_u8p),
                               seq.ctypes.data_as(cp), qs.ctypes.data_as(cp),
                               lens.ctypes.data_as(i32p), det)
    out = []
    for n in range(N):
        L = int(lens[n])
        out.append((seq[n, :L].tobytes().decode(), qs[n, :L].tobytes().decode(), moves[n].copy()))
    return out


# ---------------------------------------------------------------- f1: signal scaling (ScalerNode)
_i16p = C.POINTER(C.c_int16)
_u16p = C.POINTER(C.c_uint16)


def shift_scale_i16_to_f16(x_i16, shift, scale, use_ref=False):
    """f16((float(x) - shift) / scale) as np.float16 (tensor_utils.cpp:89-142)."""
    x = np.ascontiguousarray(x_i16, np.int16)
    out = np.empty(x.shape, np.uint16)
    fn = ref().ref_shift_scale_i16_to_f16 if use_ref else lib().orc_shift_scale_i16_to_f16
    fn(x.ctypes.data_as(_i16p), C.c_long(x.size), C.c_float(shift), C.c_float(scale),
       out.ctypes.data_as(_u16p))
    return out.view(np.float16)


def quantile_counting(x_i16, q, use_ref=False):
    x = np.ascontiguousarray(x_i16, np.int16)
    qq = np.ascontiguousarray(q, np.float32)
    out = np.empty(qq.size, np.float32)
    fn = ref().ref_quantile_counting if use_ref else lib().orc_quantile_counting
    fn(x.ctypes.data_as(_i16p), C.c_long(x.size), _fp(qq), C.c_int(qq.size), _fp(out))
    return out


def med_mad(x_i16, use_ref=False):
    x = np.ascontiguousarray(x_i16, np.int16)
    med, mad = C.c_float(), C.c_float()
    fn = ref().ref_med_mad_expr if use_ref else lib().orc_med_mad
    fn(x.ctypes.data_as(_i16p), C.c_long(x.size), C.byref(med), C.byref(mad))
    return med.value, mad.value


def quantile_shift_scale(x_i16, quantile_a, quantile_b, shift_multiplier, scale_multiplier):
    x = np.ascontiguousarray(x_i16, np.int16)
    sh, sc = C.c_float(), C.c_float()
    lib().orc_quantile_shift_scale(x.ctypes.data_as(_i16p), C.c_long(x.size), C.c_float(quantile_a),
                                   C.c_float(quantile_b), C.c_float(shift_multiplier),
                                   C.c_float(scale_multiplier), C.byref(sh), C.byref(sc))
    return sh.value, sc.value


def pa_shift_scale(scaling, offset, standardise, mean, stdev, open_pore_level=float("nan"),
                   expected_open_pore_level=0.0):
    sh, sc, adj = C.c_float(), C.c_float(), C.c_float()
    lib().orc_pa_shift_scale(C.c_float(scaling), C.c_float(offset), C.c_int(int(standardise)),
                             C.c_float(mean), C.c_float(stdev), C.c_float(open_pore_level),
                             C.c_float(expected_open_pore_level), C.byref(sh), C.byref(sc),
                             C.byref(adj))
    return sh.value, sc.value, adj.value


def trim(signal_f32, threshold=2.4, window_size=40, min_elements=3):
    s = np.ascontiguousarray(signal_f32, np.float32)
    return int(lib().orc_trim(_fp(s), C.c_int(s.size), C.c_float(threshold), C.c_int(window_size),
                              C.c_int(min_elements)))


def rna_adapter_pos(x_i16, use_ref=False):
    """ScalerNode.cpp:58-107 (determine_rna_adapter_pos); use_ref: the reference's own function (ref_scaler.cpp)."""
    x = np.ascontiguousarray(x_i16, np.int16)
    if use_ref:
        return int(ref_scaler().ref_determine_rna_adapter_pos(x.ctypes.data_as(_i16p), C.c_long(x.size)))
    return int(lib().orc_rna_adapter_pos(x.ctypes.data_as(_i16p), C.c_long(x.size)))


_ref_scaler = None
REF_SCALER_SO = os.path.join(os.path.dirname(REF_SO), "libdorado_ref_scaler.so")
SCALING_STRATEGIES = {"med_mad": 0, "quantile": 1, "pa": 2}   # config::ScalingStrategy order


def ref_scaler():
    """oracle/_ref/libdorado_ref_scaler.so: the reference's ScalerNode.cpp compiled in place + MessageSink, kits, tensor_utils
    (oracle/Makefile.ref, oracle/ref_scaler.cpp)."""
    global _ref_scaler
    if _ref_scaler is None:
        import torch  # noqa: F401  (libtorch must be loaded first)

        _ref_scaler = C.CDLL(REF_SCALER_SO)
        _ref_scaler.ref_scaler_last_error.restype = C.c_char_p
        _ref_scaler.ref_expected_open_pore_level.restype = C.c_float
    return _ref_scaler


def ref_scaler_node(raw_i16, strategy="quantile", quantile=(0.2, 0.9, 0.51, 0.53), standardisation=(False, 0.0, 1.0),
                    is_rna_model=False, scaling=1.0, offset=0.0, open_pore_level=float("nan"), flow_cell_product_code=""):
    """One read through the REFERENCE's ScalerNode (a real node with its worker thread; ref_scaler.cpp) ->
    dict(signal np.float16, scale_pa, shift_pa, num_trimmed_samples, rna_adapter_end_signal_pos)."""
    x = np.ascontiguousarray(raw_i16, np.int16)
    p7 = (C.c_float * 7)(*quantile, float(bool(standardisation[0])), standardisation[1], standardisation[2])
    cal = (C.c_float * 3)(scaling, offset, open_pore_level)
    out = np.empty(max(x.size, 1), np.uint16)
    on = C.c_long()
    f2 = (C.c_float * 2)()
    i2 = (C.c_int * 2)()
    r = ref_scaler()
    rc = r.ref_scaler_node(x.ctypes.data_as(_i16p), C.c_long(x.size), C.c_int(SCALING_STRATEGIES[strategy]), p7,
                           C.c_int(int(is_rna_model)), cal, flow_cell_product_code.encode(), out.ctypes.data_as(_u16p),
                           C.byref(on), f2, i2)
    if rc != 0:
        raise RuntimeError(r.ref_scaler_last_error().decode())
    return {"signal": out[:on.value].view(np.float16).copy(), "scale_pa": float(f2[0]), "shift_pa": float(f2[1]),
            "num_trimmed_samples": int(i2[0]), "rna_adapter_end_signal_pos": int(i2[1])}


EXPECTED_OPEN_PORE_LEVEL = {   # ScalerNode.cpp:109-127 (flow cell product code -> pA)
    "FLO-FLG114": 200.0, "FLO-FLG114HD": 200.0, "FLO-MIN004RA": 195.50, "FLO-PRO004RA": 194.97, "FLO-MIN114": 197.61,
    "FLO-MIN114HD": 197.61, "FLO-PRO114": 199.21, "FLO-PRO114HD": 199.21, "FLO-PRO114M": 199.21,
}


def scaler_node(raw_i16, strategy="quantile", quantile=(0.2, 0.9, 0.51, 0.53), standardisation=(False, 0.0, 1.0),
                is_rna_model=False, has_rna_based_adapters=False, scaling=1.0, offset=0.0, open_pore_level=float("nan"),
                flow_cell_product_code=""):
    """CPU restatement of ScalerNode::input_thread_fn for one read (ScalerNode.cpp:144-267), composed from the restated
    pieces above; same dict as ref_scaler_node."""
    x = np.ascontiguousarray(raw_i16, np.int16)
    trim_start, rna_end = 0, 0
    if is_rna_model and not has_rna_based_adapters:          # :157-184
        pos = rna_adapter_pos(x)
        if pos < x.size:
            trim_start, rna_end = pos, 0
            x = x[pos:]
        else:
            rna_end = pos
    f32 = np.float32
    adj = f32(0.0)
    if strategy == "pa":                                     # :190-215
        sh, sc, adj = pa_shift_scale(scaling, offset, standardisation[0], standardisation[1], standardisation[2],
                                     open_pore_level, EXPECTED_OPEN_PORE_LEVEL.get(flow_cell_product_code, 0.0))
    elif strategy == "quantile":                             # :216-224
        sh, sc = quantile_shift_scale(x[rna_end:], *quantile)
    else:
        sh, sc = med_mad(x[rna_end:])
    sh, sc, adj = f32(sh), f32(sc), f32(adj)
    sig = shift_scale_i16_to_f16(x, float(sh + adj), float(sc))      # :228-229
    scale_pa = f32(scaling) * sc                             # :226-227
    shift_pa = f32(scaling) * (sh + f32(offset))
    if not is_rna_model:                                     # :233-254
        if trim_start == 0 and standardisation[0]:
            trim_start = 10
        elif trim_start == 0:
            trim_start = trim(sig[:min(8000, x.size // 2)].astype(np.float32))
        if trim_start < x.size:
            sig = sig[trim_start:]
        else:
            trim_start = 0
    return {"signal": sig, "scale_pa": float(scale_pa), "shift_pa": float(shift_pa), "num_trimmed_samples": int(trim_start),
            "rna_adapter_end_signal_pos": int(rna_end)}


def trimtest_signal(n=2000):
    """The input of tests/TrimTest.cpp "Test trim signal" regenerated by the compiled driver."""
    out = np.empty(n, np.float32)
    ref().ref_trimtest_signal(_fp(out), C.c_int(n))
    return out


# ---------------------------------------------------------------- f2: POD5 VBZ (svb16 stage)
def svb16_decode(stream_u8, n):
    s = np.ascontiguousarray(stream_u8, np.uint8)
    out = np.empty(n, np.int16)
    lib().orc_svb16_decode.restype = C.c_long
    used = lib().orc_svb16_decode(s.ctypes.data_as(_u8p), C.c_long(s.size), C.c_long(n), out.ctypes.data_as(_i16p))
    return out, int(used)


def svb16_encode(x_i16):
    x = np.ascontiguousarray(x_i16, np.int16)
    buf = np.zeros((x.size + 7) // 8 + 2 * x.size + 8, np.uint8)
    lib().orc_svb16_encode.restype = C.c_long
    used = lib().orc_svb16_encode(x.ctypes.data_as(_i16p), C.c_long(x.size), buf.ctypes.data_as(_u8p))
    return buf[:used].copy()


# ---------------------------------------------------------------- f3: variable chunks
def generate_variable_chunks(num_samples, chunk_size, stride, overlap, use_ref=False):
    """-> list of (begin, end), or raises ValueError where the reference throws (chunk.cpp:49-107)."""
    cap = 1 << 17
    out = (C.c_uint64 * (2 * cap))()
    if use_ref:
        n = ref().ref_generate_variable_chunks(C.c_uint64(num_samples), C.c_uint64(chunk_size), C.c_uint64(stride),
                                               C.c_uint64(overlap), out, cap)
    else:
        lib().orc_generate_variable_chunks.restype = C.c_long
        n = lib().orc_generate_variable_chunks(C.c_uint64(num_samples), C.c_uint64(chunk_size), C.c_uint64(stride),
                                               C.c_uint64(overlap), out, C.c_long(cap))
    if n < 0:
        raise ValueError("generate_variable_chunks: invalid arguments")
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


# ---------------------------------------------------------------- the reference's whole simplex hot path on the CPU (round 5)
_ref_pipeline = None
REF_PIPELINE_SO = os.path.join(os.path.dirname(REF_SO), "libdorado_ref_pipeline.so")


def have_ref_pipeline() -> bool:
    return os.path.exists(REF_PIPELINE_SO)


def ref_pipeline_lib():
    """oracle/_ref/libdorado_ref_pipeline.so: ScalerNode.cpp -> BasecallerNode.cpp -> basecall/ModelRunner.cpp (CRFModel / TxModel,
    CPUDecoder), all compiled in place (oracle/Makefile.ref, oracle/ref_pipeline.cpp)."""
    global _ref_pipeline
    if _ref_pipeline is None:
        import torch  # noqa: F401  (libtorch must be loaded first)

        _ref_pipeline = C.CDLL(REF_PIPELINE_SO)
        _ref_pipeline.ref_pipeline_last_error.restype = C.c_char_p
    return _ref_pipeline


def ref_pipeline(cfg, weights, raws_i16, calibration, strategy="pa", quantile=(0.2, 0.9, 0.51, 0.53),
                 standardisation=(False, 0.0, 1.0), is_rna_model=False, flow_cell_product_code="", batch_size=16,
                 num_runners=4):
    """Raw int16 reads through the REFERENCE's ScalerNode -> BasecallerNode -> CPU ModelRunner.  calibration: per read (scaling,
    offset, open_pore_level).  -> list of dict(seq, qstr, moves, scale_pa, shift_pa, num_trimmed_samples,
    rna_adapter_end_signal_pos, scaled_len)."""
    raws = [np.ascontiguousarray(r, np.int16) for r in raws_i16]
    n = len(raws)
    sig = np.ascontiguousarray(np.concatenate(raws))
    lens = np.array([r.size for r in raws], np.int64)
    d = cfg.to_desc()
    ws, arr = _wptrs(weights)
    wn = np.array([w.size for w in ws], np.int64)
    p7 = (C.c_float * 7)(*quantile, float(bool(standardisation[0])), standardisation[1], standardisation[2])
    cal = np.ascontiguousarray(calibration, np.float32).reshape(n, 3)
    stride = cfg.conv_stride // cfg.tx.up_scale_factor if cfg.tx is not None else cfg.stride
    pitch = int(lens.max()) // stride + 16
    seq = np.zeros((n, pitch), np.uint8)
    qs = np.zeros((n, pitch), np.uint8)
    mv = np.zeros((n, pitch), np.uint8)
    sl, ml, scl = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)
    f2 = np.zeros((n, 2), np.float32)
    i2 = np.zeros((n, 2), np.int32)
    i64p = C.POINTER(C.c_int64)
    L = ref_pipeline_lib()
    rc = L.ref_pipeline_run(C.byref(d), arr, wn.ctypes.data_as(i64p), len(ws), C.c_float(cfg.qscale), C.c_float(cfg.qbias),
                            C.c_int(cfg.chunk_size), C.c_int(cfg.overlap), C.c_int(batch_size), C.c_int(num_runners),
                            C.c_int(SCALING_STRATEGIES[strategy]), p7, C.c_int(int(is_rna_model)),
                            sig.ctypes.data_as(_i16p), lens.ctypes.data_as(i64p), C.c_int(n), _fp(cal),
                            flow_cell_product_code.encode(), C.c_int(pitch), seq.ctypes.data_as(C.c_char_p),
                            qs.ctypes.data_as(C.c_char_p), mv.ctypes.data_as(_u8p), sl.ctypes.data_as(i64p),
                            ml.ctypes.data_as(i64p), _fp(f2), i2.ctypes.data_as(C.POINTER(C.c_int)), scl.ctypes.data_as(i64p))
    if rc != 0:
        raise RuntimeError(L.ref_pipeline_last_error().decode())
    return [{"seq": seq[r, :sl[r]].tobytes().decode(), "qstr": qs[r, :sl[r]].tobytes().decode(), "moves": mv[r, :ml[r]].copy(),
             "scale_pa": float(f2[r, 0]), "shift_pa": float(f2[r, 1]), "num_trimmed_samples": int(i2[r, 0]),
             "rna_adapter_end_signal_pos": int(i2[r, 1]), "scaled_len": int(scl[r])} for r in range(n)]
