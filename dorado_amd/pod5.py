"""POD5 reader without libpod5: the host half of the reference's DataLoader
(dorado/data_loader/DataLoader.cpp:114-260), which goes through pod5_format/c_api.h of the un-vendored
pod5-file-format 0.3.36.

File layout (pod5 format specification):
    signature "\\x8bPOD\\r\\n\\x1a\\n" | section marker (16 B uuid) |
    embedded Arrow IPC file: signal table  [read_id fixed16, signal large_binary (VBZ), samples u32]
    embedded Arrow IPC file: run-info table
    embedded Arrow IPC file: reads table   [read_id, signal list<u64> (rows of the signal table), calibration_*, ...]
    "FOOTER\\0\\0" | flatbuffer Footer {file_identifier, software, pod5_version, contents:[EmbeddedFile{offset,
    length, format, content_type}]} | footer length i64 | section marker | signature
The Arrow tables are read with pyarrow (the only Arrow available offline); a signal row is
zstd(svb16(zigzag(delta(int16)))): the zstd frame is inflated here on the host (libzstd.so.1 through ctypes,
pyarrow's bundled codec as fallback), the svb16 stage runs on the device (Engine.svb16_decode ->
mibc_svb16_decode), so a read's samples are born in HBM.

This module is host plumbing above the C-ABI; it never touches oracle/."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import struct
import uuid
from dataclasses import dataclass, field

import numpy as np

SIGNATURE = b"\x8bPOD\r\n\x1a\n"
FOOTER_MAGIC = b"FOOTER\x00\x00"
CONTENT_READS, CONTENT_SIGNAL, CONTENT_READ_ID_INDEX, CONTENT_OTHER_INDEX, CONTENT_RUN_INFO = 0, 1, 2, 3, 4


class Pod5Error(RuntimeError):
    pass


# ---------------------------------------------------------------- zstd (frame inflate, host)
_zstd = None


def _zstd_lib():
    global _zstd
    if _zstd is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        try:
            L = C.CDLL(name)
            L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
            L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
            L.ZSTD_decompress.restype = C.c_size_t
            L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.ZSTD_isError.restype = C.c_uint
            L.ZSTD_isError.argtypes = [C.c_size_t]
            _zstd = L
        except OSError:
            _zstd = False
    return _zstd


def zstd_frame_content_size(buf: bytes) -> int:
    """Frame_Content_Size of a zstd frame header (RFC 8878 §3.1.1.1)."""
    if len(buf) < 6 or buf[:4] != b"\x28\xb5\x2f\xfd":
        raise Pod5Error("not a zstd frame")
    fhd = buf[4]
    fcs_flag, single, dict_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[dict_flag]
    size = (1 if single else 0, 2, 4, 8)[fcs_flag]
    if size == 0:
        raise Pod5Error("zstd frame without content size")
    if len(buf) < pos + size:
        raise Pod5Error("truncated zstd frame header")
    v = int.from_bytes(buf[pos:pos + size], "little")
    return v + 256 if size == 2 else v


def svb16_max_bytes(samples: int) -> int:
    """Largest StreamVByte-16 stream `samples` values can occupy: one control bit per value + 2 data bytes."""
    return (samples + 7) // 8 + 2 * samples


def zstd_inflate(buf: bytes, max_size: int | None = None) -> bytes:
    """Inflate one zstd frame.  max_size: upper bound the caller knows for the content (a signal row: the svb16 worst
    case of its `samples` column) — the size field of a corrupt / hostile frame is not trusted with an allocation."""
    n = zstd_frame_content_size(buf)
    if max_size is not None and n > max_size:
        raise Pod5Error(f"zstd: frame claims {n} bytes, the row can hold at most {max_size}")
    L = _zstd_lib()
    if L:
        out = C.create_string_buffer(n)
        r = L.ZSTD_decompress(out, n, buf, len(buf))
        if L.ZSTD_isError(r) or r != n:
            raise Pod5Error("zstd: corrupt signal row")
        return out.raw
    import pyarrow as pa

    try:
        return pa.decompress(buf, decompressed_size=n, codec="zstd", asbytes=True)
    except Exception as e:  # pyarrow raises ArrowIOError / OSError on corrupt frames
        raise Pod5Error(f"zstd: corrupt signal row ({e})") from None


# ---------------------------------------------------------------- footer (minimal flatbuffer reader)
def _fb_table_fields(buf, pos):
    """(vtable field offsets, table pos) of the flatbuffer table at `pos`."""
    vt = pos - struct.unpack_from("<i", buf, pos)[0]
    if vt < 0:
        raise IndexError("vtable before the buffer")
    vt_len = struct.unpack_from("<H", buf, vt)[0]
    n = (vt_len - 4) // 2
    return [struct.unpack_from("<H", buf, vt + 4 + 2 * i)[0] for i in range(n)], pos


def _fb_string(buf, table_pos, off):
    if off == 0:
        return ""
    p = table_pos + off
    p += struct.unpack_from("<I", buf, p)[0]
    ln = struct.unpack_from("<I", buf, p)[0]
    if p + 4 + ln > len(buf):
        raise IndexError("string runs past the footer")
    return bytes(buf[p + 4:p + 4 + ln]).decode()


def parse_footer(buf: bytes):
    """-> dict(file_identifier, software, pod5_version, contents=[(offset, length, format, content_type)]).
    Offsets inside the flatbuffer are not trusted: anything that points outside the footer is a Pod5Error."""
    try:
        return _parse_footer(buf)
    except (struct.error, IndexError, UnicodeDecodeError, OverflowError, MemoryError) as e:
        raise Pod5Error(f"corrupt POD5 footer ({type(e).__name__}: {e})") from None


def _parse_footer(buf: bytes):
    if len(buf) < 72 or buf[:8] != SIGNATURE or buf[-8:] != SIGNATURE:
        raise Pod5Error("bad POD5 signature")
    flen = struct.unpack_from("<q", buf, len(buf) - 32)[0]
    end = len(buf) - 32
    start = end - flen
    if flen <= 0 or start < 32 or buf[start - 8:start] != FOOTER_MAGIC:
        raise Pod5Error("POD5 footer not found")
    fb = memoryview(buf)[start:end]
    root = struct.unpack_from("<I", fb, 0)[0]
    fields, tp = _fb_table_fields(fb, root)
    fields += [0] * (4 - len(fields))
    out = {"file_identifier": _fb_string(fb, tp, fields[0]), "software": _fb_string(fb, tp, fields[1]),
           "pod5_version": _fb_string(fb, tp, fields[2]), "contents": []}
    if fields[3]:
        p = tp + fields[3]
        p += struct.unpack_from("<I", fb, p)[0]
        cnt = struct.unpack_from("<I", fb, p)[0]
        if cnt > len(fb) // 4:
            raise Pod5Error("corrupt POD5 footer (embedded-file count)")
        for i in range(cnt):
            ep = p + 4 + 4 * i
            ep += struct.unpack_from("<I", fb, ep)[0]
            ef, etp = _fb_table_fields(fb, ep)
            ef += [0] * (4 - len(ef))
            off = struct.unpack_from("<q", fb, etp + ef[0])[0] if ef[0] else 0
            ln = struct.unpack_from("<q", fb, etp + ef[1])[0] if ef[1] else 0
            fmt = struct.unpack_from("<h", fb, etp + ef[2])[0] if ef[2] else 0
            ct = struct.unpack_from("<h", fb, etp + ef[3])[0] if ef[3] else 0
            out["contents"].append((off, ln, fmt, ct))
    return out


# ---------------------------------------------------------------- reads
@dataclass
class Pod5Read:
    """The SimplexRead fields DataLoader fills (DataLoader.cpp:163-225)."""
    read_id: str
    signal_rows: list            # indices into the signal table
    num_samples: int
    scaling: float               # calibration_scale
    offset: float                # calibration_offset
    open_pore_level: float
    read_number: int
    start_sample: int
    channel: int
    mux: int                     # well
    end_reason: str
    pore_type: str
    num_minknow_events: int
    sample_rate: int
    flow_cell_product_code: str
    sequencing_kit: str
    flowcell_id: str
    run_id: str                  # protocol_run_id
    acquisition_id: str
    position_id: str
    sample_id: str
    experiment_id: str
    run_acquisition_start_time_ms: int
    filename: str = ""
    raw: np.ndarray | None = field(default=None, repr=False)

    @property
    def start_time_ms(self) -> int:      # DataLoader.cpp:174
        return (self.start_sample * 1000) // self.sample_rate

    @property
    def is_end_reason_mux_change(self) -> bool:   # :211-214
        return self.end_reason in ("mux_change", "unblock_mux_change")


def _ms(ts):
    if ts is None:
        return 0
    if hasattr(ts, "timestamp"):
        return int(round(ts.timestamp() * 1000))
    return int(ts)


class Pod5File:
    def __init__(self, path: str):
        import pyarrow as pa
        import pyarrow.ipc as ipc

        self.path = str(path)
        with open(self.path, "rb") as f:
            self._buf = f.read()
        self.footer = parse_footer(self._buf)
        tabs = {}
        for off, ln, fmt, ct in self.footer["contents"]:
            if fmt != 0:
                raise Pod5Error("embedded file is not Feather V2 / Arrow IPC")
            tabs[ct] = ipc.open_file(pa.BufferReader(memoryview(self._buf)[off:off + ln])).read_all()
        for need in (CONTENT_READS, CONTENT_SIGNAL, CONTENT_RUN_INFO):
            if need not in tabs:
                raise Pod5Error("POD5 file lacks a reads / signal / run-info table")
        self.reads_table, self.signal_table, self.run_info_table = tabs[CONTENT_READS], tabs[CONTENT_SIGNAL], tabs[CONTENT_RUN_INFO]
        self._sig_bytes = self.signal_table.column("signal")
        self._sig_samples = self.signal_table.column("samples").to_numpy()
        ri = self.run_info_table.to_pydict()
        self._run = {ri["acquisition_id"][i]: {k: ri[k][i] for k in ri} for i in range(self.run_info_table.num_rows)}

    @property
    def num_reads(self) -> int:
        return self.reads_table.num_rows

    def reads(self, allowed_read_ids=None, ignored_read_ids=()):
        """Metadata of every read (no signal yet), filtered like should_process_pod5_row (:98-112)."""
        t = self.reads_table.to_pydict()
        names = set(self.reads_table.schema.names)
        out = []
        for i in range(self.num_reads):
            rid = str(uuid.UUID(bytes=t["read_id"][i]))
            if (allowed_read_ids is not None and rid not in allowed_read_ids) or rid in ignored_read_ids:
                continue
            run = self._run[t["run_info"][i]]
            out.append(Pod5Read(
                read_id=rid, signal_rows=list(t["signal"][i]), num_samples=int(t["num_samples"][i]),
                scaling=float(t["calibration_scale"][i]), offset=float(t["calibration_offset"][i]),
                open_pore_level=float(t["open_pore_level"][i]) if "open_pore_level" in names else float("nan"),
                read_number=int(t["read_number"][i]), start_sample=int(t["start"][i]), channel=int(t["channel"][i]),
                mux=int(t["well"][i]), end_reason=str(t["end_reason"][i]), pore_type=str(t["pore_type"][i]),
                num_minknow_events=int(t["num_minknow_events"][i]), sample_rate=int(run["sample_rate"]),
                flow_cell_product_code=str(run["flow_cell_product_code"]), sequencing_kit=str(run["sequencing_kit"]),
                flowcell_id=str(run["flow_cell_id"]), run_id=str(run["protocol_run_id"]),
                acquisition_id=str(run["acquisition_id"]), position_id=str(run["sequencer_position"]),
                sample_id=str(run["sample_id"]), experiment_id=str(run["experiment_name"]),
                run_acquisition_start_time_ms=_ms(run["acquisition_start_time"]),
                filename=self.path.rsplit("/", 1)[-1]))
        return out

    def inflated_rows(self, rows):
        """zstd stage of the given signal-table rows -> (list of svb16 streams, samples per row)."""
        ns = [int(self._sig_samples[int(r)]) for r in rows]
        streams = [zstd_inflate(self._sig_bytes[int(r)].as_py(), svb16_max_bytes(n)) for r, n in zip(rows, ns)]
        return streams, ns

    def load_signals(self, reads, engine):
        """Fill read.raw (int16) for every read: zstd on the host, svb16 + zig-zag + delta on the device
        (pod5_get_read_complete_signal, DataLoader.cpp:163-170).  Raises on corrupt rows / length mismatch."""
        rows, owner = [], []
        for k, r in enumerate(reads):
            for row in r.signal_rows:
                rows.append(row)
                owner.append(k)
        streams, ns = self.inflated_rows(rows)
        decoded, status = engine.svb16_decode(streams, ns)
        if len(status) and status.any():
            raise Pod5Error(f"corrupt signal rows: {np.nonzero(status)[0].tolist()}")
        parts = [[] for _ in reads]
        for k, d in zip(owner, decoded):
            parts[k].append(d)
        for r, p in zip(reads, parts):
            r.raw = np.concatenate(p) if p else np.zeros(0, np.int16)
            if r.raw.size != r.num_samples:
                raise Pod5Error(f"read {r.read_id}: {r.raw.size} samples decoded, {r.num_samples} expected")
        return reads
