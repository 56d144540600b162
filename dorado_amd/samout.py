"""Unaligned SAM records with the reference's read tags (output half): what
ReadCommon::generate_read_tags / extract_sam_lines (dorado/read_pipeline/base/messages.cpp:43-121,
340-357) and utils::mean_qscore_from_qstring (dorado/utils/sequence_utils.cpp:169-191) put on a simplex
read.  Host plumbing above the C-ABI; BAM encoding itself stays with htslib in the reference."""
from __future__ import annotations

import datetime

import numpy as np

_Q_TABLE = np.zeros(256, np.float32)
for _q in range(33, 128):
    _Q_TABLE[_q] = np.float32(10.0) ** (np.float32(-(_q - 33)) / np.float32(10.0))


def mean_qscore_from_qstring(qstring: str) -> float:
    """sequence_utils.cpp:169-191: mean error probability -> phred, clamped to [1, 50]; f32, summed in order."""
    if not qstring:
        return 0.0
    e = _Q_TABLE[np.frombuffer(qstring.encode("latin1"), np.uint8)]
    total = np.float32(0.0)
    for v in e:                                   # sequential f32 accumulation, like std::accumulate
        total = np.float32(total + v)
    mean_error = np.float32(total / np.float32(len(qstring)))
    q = np.float32(-10.0) * np.log10(mean_error, dtype=np.float32)
    return float(min(max(q, np.float32(1.0)), np.float32(50.0)))


def calculate_mean_qscore(qstring: str, mean_qscore_start_pos: int = 60) -> float:
    """messages.cpp:340-357 (DNA branch)."""
    if len(qstring) <= mean_qscore_start_pos:
        return mean_qscore_from_qstring(qstring)
    return mean_qscore_from_qstring(qstring[mean_qscore_start_pos:])


def timestamp_from_unix_ms(ms: int) -> str:
    """utils::get_string_timestamp_from_unix_time_ms: 2017-09-12T09:50:12.456+00:00"""
    t = datetime.datetime.fromtimestamp(ms / 1000.0, datetime.timezone.utc)
    return t.strftime("%Y-%m-%dT%H:%M:%S.") + f"{int(ms % 1000):03d}+00:00"


def _f(x: float) -> str:
    return np.format_float_positional(np.float32(x), unique=True, trim="0") if np.isfinite(x) else "nan"


def sam_record(read_id: str, seq: str, qstring: str, moves=None, *, model_stride: int = 6, num_samples: int = 0,
               num_trimmed_samples: int = 0, sample_rate: int = 5000, mux: int = 0, channel: int = 0,
               start_time: str = "", read_number: int = 0, filename: str = "", shift_pa: float = 0.0,
               scale_pa: float = 1.0, scaling_method: str = "pa", read_group: str = "",
               mean_qscore_start_pos: int = 60) -> str:
    """One unaligned SAM line (flag 4) with the tag set and order of generate_read_tags.
    num_samples = raw samples AFTER the front trim (get_raw_data_samples()); ns / du add the trim back."""
    ns = num_samples + num_trimmed_samples
    tags = [
        f"qs:f:{_f(calculate_mean_qscore(qstring, mean_qscore_start_pos))}",
        f"du:f:{_f(np.float32(ns) / np.float32(sample_rate))}",
        f"ns:i:{ns}", f"ts:i:{num_trimmed_samples}", f"mx:i:{mux}", f"ch:i:{channel}", f"st:Z:{start_time}",
        f"rn:i:{read_number}", f"fn:Z:{filename}", f"sm:f:{_f(shift_pa)}", f"sd:f:{_f(scale_pa)}",
        f"sv:Z:{scaling_method}", "dx:i:0",
    ]
    if read_group:
        tags.append(f"RG:Z:{read_group}")
    if moves is not None:   # mv:B:c,<stride>,<moves...>  (messages.cpp:109-118)
        tags.append("mv:B:c," + ",".join([str(int(model_stride))] + [str(int(m)) for m in moves]))
    return "\t".join([read_id, "4", "*", "0", "0", "*", "*", "0", "0", seq or "*", qstring or "*"] + tags)
