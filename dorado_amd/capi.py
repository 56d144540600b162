"""ctypes binding of the C-ABI in include/mibc.h (dorado_amd/libmibc.so).

This is the ONLY way Python reaches the kernels; there is no CPU fallback — if the HIP library
is missing or no GPU is visible, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .config import ModelConfig, ModelDescC

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmibc.so")

MIBC_OK = 0
MIBC_NOT_SUPPORTED = 1


class MibcError(RuntimeError):
    pass


class MibcNotSupported(MibcError):
    pass


class DecodeOptsC(C.Structure):
    _fields_ = [("beam_width", C.c_int), ("beam_cut", C.c_float), ("blank_score", C.c_float),
                ("q_shift", C.c_float), ("q_scale", C.c_float)]


class VarChunkC(C.Structure):
    _fields_ = [("row", C.c_int), ("sample_start", C.c_int), ("n_samples", C.c_int)]


class StageMsC(C.Structure):
    _fields_ = [("conv", C.c_float), ("lstm", C.c_float), ("head", C.c_float),
                ("decode", C.c_float), ("total", C.c_float), ("lstm_layer", C.c_float * 8),
                ("h2d", C.c_float), ("d2h", C.c_float)]


EXPORTS = [
    "mibc_device_count", "mibc_device_memory", "mibc_last_error", "mibc_build_id", "mibc_create", "mibc_destroy", "mibc_query_memory",
    "mibc_reserve", "mibc_output_steps", "mibc_batch_granularity", "mibc_host_alloc",
    "mibc_host_free", "mibc_device_alloc", "mibc_device_free", "mibc_memcpy_h2d",
    "mibc_memcpy_d2h", "mibc_forward", "mibc_decode", "mibc_call_device", "mibc_call",
    "mibc_sync", "mibc_set_decode_overlap", "mibc_quantize_lstm_weights", "mibc_time_forward", "mibc_get_stage_ms", "mibc_get_stage_ms_prev", "mibc_set_profile", "mibc_debug_tap",
    "mibc_forward_i16", "mibc_call_device_i16", "mibc_call_i16", "mibc_scaler_stats", "mibc_scale_reads",
    "mibc_svb16_decode", "mibc_forward_var", "mibc_call_device_var", "mibc_call_var",
    "mibc_call_async", "mibc_call_wait", "mibc_call_poll", "mibc_call_var_async",
]

SCALE_QUANTILE = 0
SCALE_MED_MAD = 1


DBG_LIB_PATH = os.path.join(HERE, "libmibc_dbg.so")


def build() -> None:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU): the product library libmibc.so (exports
    exactly include/mibc.h) and the debug library libmibc_dbg.so (same sources + the mibc_debug_* test hooks and the
    ablation kernels; only tests/ and tools/ load it)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "csrc"), "all", "debug"])


_dbg_lib = None


def dbg_lib():
    """libmibc_dbg.so: the kernel-vs-kernel comparison / timing hooks (mibc_debug_*).  Test infrastructure only."""
    global _dbg_lib
    if _dbg_lib is None:
        if not os.path.exists(DBG_LIB_PATH):
            raise MibcError(f"{DBG_LIB_PATH} is missing: run `make -C dorado_amd/csrc debug`")
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        _dbg_lib = C.CDLL(DBG_LIB_PATH)
    return _dbg_lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MibcError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                            "g.build()'` (there is no CPU fallback)")
        try:
            import torch  # noqa: F401  share torch's libamdhip64 (same SONAME) when it is around
        except Exception:  # pragma: no cover
            pass
        L = C.CDLL(LIB_PATH)
        L.mibc_last_error.restype = C.c_char_p
        L.mibc_last_error.argtypes = [C.c_void_p]
        L.mibc_build_id.restype = C.c_char_p
        L.mibc_create.argtypes = [C.c_int, C.POINTER(ModelDescC), C.POINTER(C.POINTER(C.c_float)),
                                  C.c_int, C.POINTER(C.c_void_p)]
        L.mibc_destroy.argtypes = [C.c_void_p]
        L.mibc_destroy.restype = None
        L.mibc_query_memory.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t)]
        L.mibc_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mibc_output_steps.argtypes = [C.c_void_p, C.c_int]
        L.mibc_batch_granularity.argtypes = [C.c_void_p]
        L.mibc_host_alloc.restype = C.c_void_p
        L.mibc_host_alloc.argtypes = [C.c_size_t]
        L.mibc_host_free.argtypes = [C.c_void_p]
        L.mibc_host_free.restype = None
        L.mibc_device_alloc.restype = C.c_void_p
        L.mibc_device_alloc.argtypes = [C.c_void_p, C.c_size_t]
        L.mibc_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.mibc_device_free.restype = None
        L.mibc_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.mibc_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.mibc_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mibc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(DecodeOptsC),
                                  C.c_void_p]
        L.mibc_call_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.POINTER(DecodeOptsC), C.c_void_p]
        L.mibc_call.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(DecodeOptsC),
                                C.c_void_p]
        L.mibc_sync.argtypes = [C.c_void_p]
        L.mibc_time_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.mibc_get_stage_ms.argtypes = [C.c_void_p, C.POINTER(StageMsC)]
        L.mibc_get_stage_ms_prev.argtypes = [C.c_void_p, C.POINTER(StageMsC)]
        L.mibc_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.mibc_quantize_lstm_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mibc_debug_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.mibc_forward_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mibc_call_device_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.POINTER(DecodeOptsC), C.c_void_p]
        L.mibc_call_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.POINTER(DecodeOptsC), C.c_void_p]
        L.mibc_scaler_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.mibc_scale_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mibc_svb16_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p]
        L.mibc_forward_var.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p]
        L.mibc_call_device_var.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.POINTER(DecodeOptsC), C.c_void_p]
        L.mibc_call_var.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.POINTER(DecodeOptsC), C.c_void_p]
        L.mibc_call_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.POINTER(DecodeOptsC), C.c_void_p]
        L.mibc_call_wait.argtypes = [C.c_void_p, C.c_int]
        L.mibc_call_poll.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def device_count() -> int:
    return int(lib().mibc_device_count())


def device_memory(device: int = 0):
    """(free, total) bytes of a device (mibc_device_memory)."""
    a, b = C.c_size_t(), C.c_size_t()
    L = lib()
    L.mibc_device_memory.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    if L.mibc_device_memory(device, C.byref(a), C.byref(b)) != MIBC_OK:
        raise MibcError("mibc_device_memory failed")
    return int(a.value), int(b.value)


class Engine:
    """One engine == one device == one HIP stream (the reference's CudaCaller device half)."""

    def __init__(self, cfg: ModelConfig, weights, device: int = 0, taps: bool = False):
        L = lib()
        if device_count() <= device:
            raise MibcError(f"no HIP device {device} visible (device_count={device_count()}); "
                            "the HIP path has no CPU fallback")
        if taps:
            os.environ["MIBC_TAPS"] = "1"
        self.cfg = cfg
        self._desc = cfg.to_desc()
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        arr = (C.POINTER(C.c_float) * len(ws))(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in ws])
        h = C.c_void_p()
        rc = L.mibc_create(device, C.byref(self._desc), arr, len(ws), C.byref(h))
        if taps:
            os.environ.pop("MIBC_TAPS", None)
        if rc != MIBC_OK:
            msg = L.mibc_last_error(None).decode()
            raise (MibcNotSupported if rc > 0 else MibcError)(f"mibc_create: {msg}")
        self._h = h
        self.opts = DecodeOptsC(32, 100.0, 2.0, cfg.qbias, cfg.qscale)

    # -- helpers
    def _check(self, rc, what):
        if rc != MIBC_OK:
            msg = lib().mibc_last_error(self._h).decode()
            raise (MibcNotSupported if rc > 0 else MibcError)(f"{what}: {msg}")

    def close(self):
        if getattr(self, "_h", None):
            lib().mibc_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def batch_granularity(self) -> int:
        return int(lib().mibc_batch_granularity(self._h))

    def output_steps(self, t_in: int) -> int:
        return int(lib().mibc_output_steps(self._h, t_in))

    def reserve(self, n_max: int, t_in: int):
        self._check(lib().mibc_reserve(self._h, n_max, t_in), "mibc_reserve")

    def query_memory(self, t_in: int):
        a, b = C.c_size_t(), C.c_size_t()
        self._check(lib().mibc_query_memory(self._h, t_in, C.byref(a), C.byref(b)), "mibc_query_memory")
        return int(a.value), int(b.value)

    def device_alloc(self, nbytes: int) -> int:
        p = lib().mibc_device_alloc(self._h, nbytes)
        if not p:
            raise MibcError(f"mibc_device_alloc({nbytes}) failed")
        return p

    def device_free(self, p: int):
        lib().mibc_device_free(self._h, p)

    def h2d(self, dst: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(lib().mibc_memcpy_h2d(self._h, dst, arr.ctypes.data, arr.nbytes), "h2d")

    def d2h(self, arr: np.ndarray, src: int):
        self._check(lib().mibc_memcpy_d2h(self._h, arr.ctypes.data, src, arr.nbytes), "d2h")

    def sync(self):
        self._check(lib().mibc_sync(self._h), "mibc_sync")

    def set_profile(self, level: int):
        lib().mibc_set_profile(self._h, level)

    def set_decode_overlap(self, on: bool = True):
        """Decoder on its own stream, scores double-buffered: the decoder of a batch runs under the network of the next."""
        self._check(lib().mibc_set_decode_overlap(self._h, C.c_int(int(on))), "mibc_set_decode_overlap")

    def stage_ms(self, prev: bool = False) -> dict:
        """HIP-event stage times of the last profiled call; prev=True: of the one before it (the host can read call i - 1
        while call i is already enqueued — no pipeline bubble)."""
        s = StageMsC()
        if prev:
            self._check(lib().mibc_get_stage_ms_prev(self._h, C.byref(s)), "mibc_get_stage_ms_prev")
        else:
            self._check(lib().mibc_get_stage_ms(self._h, C.byref(s)), "mibc_get_stage_ms")
        return {"conv": s.conv, "lstm": s.lstm, "head": s.head, "decode": s.decode,
                "total": s.total, "lstm_layer": [s.lstm_layer[i] for i in range(8)]}

    def time_forward(self, n: int, t_in: int) -> float:
        ms = C.c_float()
        self._check(lib().mibc_time_forward(self._h, n, t_in, C.byref(ms)), "mibc_time_forward")
        return float(ms.value)

    def tap(self, tap: int, shape, dtype) -> np.ndarray:
        out = np.zeros(shape, dtype)
        self._check(lib().mibc_debug_tap(self._h, tap, out.ctypes.data, out.nbytes), "mibc_debug_tap")
        return out

    # -- the hot path (device pointers)
    def forward_device(self, in_dev: int, n: int, t_in: int, scores_dev: int):
        self._check(lib().mibc_forward(self._h, in_dev, n, t_in, scores_dev), "mibc_forward")

    def decode_device(self, scores_dev: int, n: int, t: int, out_dev: int):
        self._check(lib().mibc_decode(self._h, scores_dev, n, t, C.byref(self.opts), out_dev),
                    "mibc_decode")

    def call_device(self, in_dev: int, n: int, t_in: int, out_dev: int):
        self._check(lib().mibc_call_device(self._h, in_dev, n, t_in, C.byref(self.opts), out_dev),
                    "mibc_call_device")

    # -- numpy conveniences (tests / small runs)
    def forward(self, x_f16: np.ndarray) -> np.ndarray:
        """x [N, T_in] f16 -> scores [N, T, K] f16."""
        x = np.ascontiguousarray(x_f16, np.float16)
        n, t_in = x.shape
        t = self.output_steps(t_in)
        k = self.cfg.outsize
        self.reserve(n, t_in)
        d_in = self.device_alloc(x.nbytes)
        d_sc = self.device_alloc(n * t * k * 2)
        try:
            self.h2d(d_in, x)
            self.forward_device(d_in, n, t_in, d_sc)
            self.sync()
            out = np.zeros((n, t, k), np.float16)
            self.d2h(out, d_sc)
        finally:
            self.device_free(d_in)
            self.device_free(d_sc)
        return out

    def decode(self, scores_f16: np.ndarray, t_in_for_reserve: int | None = None):
        """scores [N, T, K] f16 -> list of (seq, qstr, moves)."""
        s = np.ascontiguousarray(scores_f16, np.float16)
        n, t, k = s.shape
        t_in = t_in_for_reserve if t_in_for_reserve is not None else t * self.cfg.stride
        assert self.output_steps(t_in) == t
        g = int(lib().mibc_batch_granularity(self._h))
        self.reserve(max(g, (n + g - 1) // g * g), t_in)
        d_sc = self.device_alloc(s.nbytes)
        d_out = self.device_alloc(3 * n * t)
        try:
            self.h2d(d_sc, s)
            self.decode_device(d_sc, n, t, d_out)
            self.sync()
            out = np.zeros((3, n, t), np.int8)
            self.d2h(out, d_out)
        finally:
            self.device_free(d_sc)
            self.device_free(d_out)
        return unpack_planes(out)

    def call(self, x_f16: np.ndarray):
        """Host batch -> decoded chunks, through mibc_call (H2D + forward + decode + D2H)."""
        x = np.ascontiguousarray(x_f16, np.float16)
        n, t_in = x.shape
        t = self.output_steps(t_in)
        out = np.zeros((3, n, t), np.int8)
        self._check(lib().mibc_call(self._h, x.ctypes.data, n, t_in, C.byref(self.opts),
                                    out.ctypes.data), "mibc_call")
        return unpack_planes(out)

    def call_two_slots(self, batches):
        """Several host batches through the two-phase entry points (mibc_call_async / mibc_call_wait), two in
        flight, pinned buffers.  Returns the decoded chunks per batch."""
        L = lib()
        xs = [np.ascontiguousarray(b, np.float16) for b in batches]
        n, t_in = xs[0].shape
        t = self.output_steps(t_in)
        self.reserve(n, t_in)
        pin_in = [L.mibc_host_alloc(n * t_in * 2) for _ in range(2)]
        pin_out = [L.mibc_host_alloc(3 * n * t) for _ in range(2)]
        res = [None] * len(xs)
        try:
            def submit(i):
                s = i & 1
                C.memmove(pin_in[s], xs[i].ctypes.data, xs[i].nbytes)
                self._check(L.mibc_call_async(self._h, s, pin_in[s], None, n, t_in, C.byref(self.opts), pin_out[s]),
                            "mibc_call_async")

            def collect(i):
                s = i & 1
                self._check(L.mibc_call_wait(self._h, s), "mibc_call_wait")
                out = np.frombuffer((C.c_int8 * (3 * n * t)).from_address(pin_out[s]), np.int8).reshape(3, n, t).copy()
                res[i] = unpack_planes(out)

            for i in range(len(xs)):
                if i >= 2:
                    collect(i - 2)
                submit(i)
            for i in range(max(0, len(xs) - 2), len(xs)):
                collect(i)
        finally:
            for p in pin_in + pin_out:
                L.mibc_host_free(p)
        return res

    def call_two_slots_mixed(self, batches):
        """batches: list of (x_rows [N, T_in] f16, chunks) with chunks = None (fixed batch: mibc_call_async) or
        [(row, sample_start, n_samples)] (variable batch: mibc_call_var_async), alternating over the two slots, two in
        flight.  Returns per batch the raw output planes int8 [3][N][T]."""
        L = lib()
        L.mibc_call_var_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.POINTER(DecodeOptsC), C.c_void_p]
        xs = [np.ascontiguousarray(b[0], np.float16) for b in batches]
        n, t_in = xs[0].shape
        t = self.output_steps(t_in)
        self.reserve(n, t_in)
        pin_in = [L.mibc_host_alloc(n * t_in * 2) for _ in range(2)]
        pin_out = [L.mibc_host_alloc(3 * n * t) for _ in range(2)]
        res = [None] * len(xs)
        try:
            def submit(i):
                s = i & 1
                C.memmove(pin_in[s], xs[i].ctypes.data, xs[i].nbytes)
                ch = batches[i][1]
                if ch is None:
                    self._check(L.mibc_call_async(self._h, s, pin_in[s], None, n, t_in, C.byref(self.opts), pin_out[s]),
                                "mibc_call_async")
                else:
                    arr = self._var_chunks(ch)      # consumed before the call returns
                    self._check(L.mibc_call_var_async(self._h, s, pin_in[s], None, n, t_in, arr, len(ch), C.byref(self.opts),
                                                      pin_out[s]), "mibc_call_var_async")

            def collect(i):
                s = i & 1
                self._check(L.mibc_call_wait(self._h, s), "mibc_call_wait")
                res[i] = np.frombuffer((C.c_int8 * (3 * n * t)).from_address(pin_out[s]), np.int8).reshape(3, n, t).copy()

            for i in range(len(xs)):
                if i >= 2:
                    collect(i - 2)
                submit(i)
            for i in range(max(0, len(xs) - 2), len(xs)):
                collect(i)
        finally:
            for p in pin_in + pin_out:
                L.mibc_host_free(p)
        return res

    # -- f1: ScalerNode on the device (raw int16 in)
    def call_i16(self, x_i16: np.ndarray, shift_scale: np.ndarray):
        """Host batch of RAW int16 chunks + one (shift, scale) per chunk -> decoded chunks
        (mibc_call_i16: the ScalerNode map is applied inside conv1's input read)."""
        x = np.ascontiguousarray(x_i16, np.int16)
        ss = np.ascontiguousarray(shift_scale, np.float32).reshape(-1, 2)
        n, t_in = x.shape
        assert ss.shape[0] == n
        t = self.output_steps(t_in)
        self.reserve(n, t_in)
        out = np.zeros((3, n, t), np.int8)
        self._check(lib().mibc_call_i16(self._h, x.ctypes.data, ss.ctypes.data, n, t_in,
                                        C.byref(self.opts), out.ctypes.data), "mibc_call_i16")
        return unpack_planes(out)

    def forward_i16(self, x_i16: np.ndarray, shift_scale: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x_i16, np.int16)
        ss = np.ascontiguousarray(shift_scale, np.float32).reshape(-1, 2)
        n, t_in = x.shape
        t = self.output_steps(t_in)
        k = self.cfg.outsize
        self.reserve(n, t_in)
        d_in, d_ss = self.device_alloc(x.nbytes), self.device_alloc(ss.nbytes)
        d_sc = self.device_alloc(n * t * k * 2)
        try:
            self.h2d(d_in, x)
            self.h2d(d_ss, ss)
            self._check(lib().mibc_forward_i16(self._h, d_in, d_ss, n, t_in, d_sc), "mibc_forward_i16")
            self.sync()
            out = np.zeros((n, t, k), np.float16)
            self.d2h(out, d_sc)
        finally:
            for p in (d_in, d_ss, d_sc):
                self.device_free(p)
        return out

    def scaler_stats(self, reads, strategy: int, params4=None):
        """reads: list of int16 arrays -> (shift_scale [n,2] f32, raw [n,2] f32) computed on the device
        (mibc_scaler_stats)."""
        sig = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int16) for r in reads]) if reads
                                   else np.zeros(0, np.int16))
        off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
        n = len(reads)
        d_sig = self.device_alloc(max(sig.nbytes, 16))
        d_off = self.device_alloc(off.nbytes)
        d_ss = self.device_alloc(max(n, 1) * 8)
        d_raw = self.device_alloc(max(n, 1) * 8)
        try:
            if sig.nbytes:
                self.h2d(d_sig, sig)
            self.h2d(d_off, off)
            p = None
            if params4 is not None:
                p = (C.c_float * 4)(*[float(v) for v in params4])
            self._check(lib().mibc_scaler_stats(self._h, d_sig, d_off, n, strategy, p, d_ss, d_raw),
                        "mibc_scaler_stats")
            self.sync()
            ss = np.zeros((n, 2), np.float32)
            raw = np.zeros((n, 2), np.float32)
            if n:
                self.d2h(ss, d_ss)
                self.d2h(raw, d_raw)
        finally:
            for q in (d_sig, d_off, d_ss, d_raw):
                self.device_free(q)
        return ss, raw

    def scale_reads(self, reads, shift_scale: np.ndarray):
        """list of int16 reads + [n,2] (shift, scale) -> list of np.float16 arrays (mibc_scale_reads)."""
        sig = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int16) for r in reads]))
        off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
        ss = np.ascontiguousarray(shift_scale, np.float32).reshape(-1, 2)
        n = len(reads)
        d_sig, d_off = self.device_alloc(max(sig.nbytes, 16)), self.device_alloc(off.nbytes)
        d_ss, d_out = self.device_alloc(ss.nbytes), self.device_alloc(max(sig.nbytes, 16))
        try:
            self.h2d(d_sig, sig)
            self.h2d(d_off, off)
            self.h2d(d_ss, ss)
            self._check(lib().mibc_scale_reads(self._h, d_sig, d_off, n, d_ss, d_out), "mibc_scale_reads")
            self.sync()
            out = np.zeros(sig.shape, np.float16)
            self.d2h(out, d_out)
        finally:
            for q in (d_sig, d_off, d_ss, d_out):
                self.device_free(q)
        return [out[off[i]:off[i + 1]] for i in range(n)]

    # -- f3: variable chunk sizes (several chunks per batch row)
    @staticmethod
    def _var_chunks(chunks):
        arr = (VarChunkC * len(chunks))()
        for i, (row, s0, n) in enumerate(chunks):
            arr[i] = VarChunkC(int(row), int(s0), int(n))
        return arr

    def forward_var(self, x_rows: np.ndarray, chunks, shift_scale=None) -> np.ndarray:
        """x_rows [N, T_in] (f16, or int16 with shift_scale [N,2]); chunks = [(row, sample_start, n_samples)].
        Returns the packed scores [N, T, K] f16 (gaps hold garbage)."""
        raw = shift_scale is not None
        x = np.ascontiguousarray(x_rows, np.int16 if raw else np.float16)
        n, t_in = x.shape
        t, k = self.output_steps(t_in), self.cfg.outsize
        self.reserve(n, t_in)
        d_in, d_sc = self.device_alloc(x.nbytes), self.device_alloc(n * t * k * 2)
        d_ss = None
        try:
            self.h2d(d_in, x)
            if raw:
                ss = np.ascontiguousarray(shift_scale, np.float32).reshape(n, 2)
                d_ss = self.device_alloc(ss.nbytes)
                self.h2d(d_ss, ss)
            arr = self._var_chunks(chunks)
            self._check(lib().mibc_forward_var(self._h, d_in, d_ss, n, t_in, arr, len(chunks), d_sc),
                        "mibc_forward_var")
            self.sync()
            out = np.zeros((n, t, k), np.float16)
            self.d2h(out, d_sc)
        finally:
            for q in (d_in, d_sc, d_ss):
                if q:
                    self.device_free(q)
        return out

    def call_var(self, x_rows: np.ndarray, chunks, shift_scale=None):
        """-> list of (seq, qstr, moves[T_c]) per chunk, through mibc_call_var."""
        raw = shift_scale is not None
        x = np.ascontiguousarray(x_rows, np.int16 if raw else np.float16)
        n, t_in = x.shape
        t = self.output_steps(t_in)
        self.reserve(n, t_in)
        ss = np.ascontiguousarray(shift_scale, np.float32).reshape(n, 2) if raw else None
        out = np.zeros((3, n, t), np.int8)
        arr = self._var_chunks(chunks)
        self._check(lib().mibc_call_var(self._h, x.ctypes.data, ss.ctypes.data if raw else None, n, t_in, arr,
                                        len(chunks), C.byref(self.opts), out.ctypes.data), "mibc_call_var")
        stride = t_in // t
        res = []
        for row, s0, ns in chunks:
            t0, tc = s0 // stride, ns // stride
            mv = out[0, row, t0:t0 + tc].astype(np.uint8)
            nb = int(mv.sum())
            res.append((out[1, row, t0:t0 + nb].tobytes().decode("ascii"),
                        out[2, row, t0:t0 + nb].tobytes().decode("ascii"), mv))
        return res

    # -- f2: POD5 VBZ (svb16 stage) on the device
    def svb16_decode(self, streams, n_samples):
        """streams: list of inflated (post-zstd) signal rows (bytes / uint8 arrays); n_samples: values per row.
        Returns (list of int16 arrays, status int32[n_rows]) decoded by mibc_svb16_decode."""
        rows = [np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray, memoryview)) else
                np.ascontiguousarray(s, np.uint8) for s in streams]
        n = len(rows)
        soff = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        noff = np.concatenate([[0], np.cumsum(n_samples)]).astype(np.int64)
        blob = np.concatenate(rows) if n else np.zeros(0, np.uint8)
        d_b = self.device_alloc(max(blob.nbytes, 16))
        d_so, d_no = self.device_alloc(soff.nbytes), self.device_alloc(noff.nbytes)
        d_out = self.device_alloc(max(int(noff[-1]) * 2, 16))
        d_st = self.device_alloc(max(n, 1) * 4)
        try:
            if blob.nbytes:
                self.h2d(d_b, blob)
            self.h2d(d_so, soff)
            self.h2d(d_no, noff)
            self._check(lib().mibc_svb16_decode(self._h, d_b, d_so, d_no, n, d_out, d_st), "mibc_svb16_decode")
            self.sync()
            out = np.zeros(int(noff[-1]), np.int16)
            st = np.zeros(n, np.int32)
            if out.size:
                self.d2h(out, d_out)
            if n:
                self.d2h(st, d_st)
        finally:
            for q in (d_b, d_so, d_no, d_out, d_st):
                self.device_free(q)
        return [out[noff[i]:noff[i + 1]] for i in range(n)], st


def unpack_planes(out3: np.ndarray):
    """int8 [3][N][T] (moves | bases | qstring) -> [(seq, qstr, moves[T] u8)], the slicing that
    CUDADecoder::beam_search_part_2 does (decode/CUDADecoder.cpp:115-173)."""
    moves, seq, qs = out3[0], out3[1], out3[2]
    res = []
    for i in range(moves.shape[0]):
        nb = int(moves[i].sum())
        res.append((seq[i, :nb].tobytes().decode("ascii"), qs[i, :nb].tobytes().decode("ascii"),
                    moves[i].astype(np.uint8)))
    return res


def quantize_lstm_weights(w_ih: np.ndarray, w_hh: np.ndarray):
    """Host-only: the lstm_quant weight quantisation (mibc_quantize_lstm_weights) -> (int8 [4C][2C], f32 scale [4C])."""
    C4, Cc = w_ih.shape
    assert C4 == 4 * Cc and w_hh.shape == (C4, Cc)
    a = np.ascontiguousarray(w_ih, np.float32)
    b = np.ascontiguousarray(w_hh, np.float32)
    q = np.zeros((C4, 2 * Cc), np.int8)
    sc = np.zeros(C4, np.float32)
    rc = lib().mibc_quantize_lstm_weights(a.ctypes.data, b.ctypes.data, Cc, q.ctypes.data, sc.ctypes.data)
    if rc != 0:
        raise MibcError(f"mibc_quantize_lstm_weights: status {rc}")
    return q, sc
