#!/usr/bin/env python3
"""tests/golden/scaler_node.npz: raw int16 reads run through the REFERENCE's ScalerNode (oracle/_ref/libdorado_ref_scaler.so =
read_pipeline/nodes/ScalerNode.cpp compiled in place, a real node fed through push_message — oracle/ref_scaler.cpp), for DNA and
RNA004 models and the three scaling strategies.  Per case: the CRC-32 of the scaled + trimmed f16 signal (bit-exact contract), its
first 128 values, its length, read_common.scale / shift (pA), num_trimmed_samples, rna_adapter_end_signal_pos.  Needs /root/reference (oracle/Makefile.ref); the fixture travels.
    python tests/golden/make_golden_scaler_node.py"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_py as O  # noqa: E402


def reads():
    rng = np.random.default_rng(20260925)

    def seg(n, level, noise):
        return rng.normal(level, noise, n)

    def i16(x):
        return np.clip(np.round(x), -32768, 32767).astype(np.int16)

    out = []
    # DNA-like: an open-pore / adapter peak in front (what utils::trim looks for), then signal
    out.append(("dna_peak", i16(np.concatenate([seg(40, 900, 20), seg(140, 1250, 40), seg(3600, 480, 85)]))))
    out.append(("dna_long_peak", i16(np.concatenate([seg(700, 1300, 30), seg(3000, 500, 90)]))))
    out.append(("dna_no_peak", i16(seg(3000, 470, 80))))
    out.append(("dna_short", i16(seg(15, 500, 60))))                 # 10 of its 15 samples go
    out.append(("dna_tiny", i16(seg(8, 500, 60))))                   # the trim would swallow the read -> 0
    out.append(("dna_outliers", i16(np.concatenate([seg(1500, 450, 70), [32767, -32768, 3000, -500], seg(1500, 450, 70)]))))
    # dRNA-like: DNA adapter at a low level, then the RNA signal higher (what determine_rna_adapter_pos looks for)
    out.append(("rna_adapter", i16(np.concatenate([seg(2200, 480, 30), seg(3600, 830, 95)]))))
    out.append(("rna_adapter_ramp", i16(np.concatenate([seg(1800, 520, 25), np.linspace(520, 700, 300) + seg(300, 0, 25),
                                                         seg(3500, 705, 80)]))))
    out.append(("rna_no_jump", i16(seg(4000, 600, 60))))             # no adapter found: nothing is cut
    out.append(("rna_late_jump", i16(np.concatenate([seg(4400, 500, 30), seg(1200, 900, 90)]))))   # behind 3n/4: not found
    out.append(("rna_short", i16(seg(1200, 600, 50))))               # shorter than the search start
    return out


CONFIGS = [  # (name, strategy, quantile params, (standardise, mean, stdev), scaling, offset, open_pore_level, flow cell)
    ("quantile", "quantile", (0.2, 0.9, 0.51, 0.53), (False, 0.0, 1.0), 0.1755, -243.0, float("nan"), ""),
    ("med_mad", "med_mad", (0.2, 0.9, 0.51, 0.53), (False, 0.0, 1.0), 0.1755, -243.0, float("nan"), ""),
    ("pa_std", "pa", (0.2, 0.9, 0.51, 0.53), (True, 93.69, 23.51), 0.1462, -228.0, 204.7, "FLO-PRO114M"),
    ("pa_raw", "pa", (0.2, 0.9, 0.51, 0.53), (False, 0.0, 1.0), 0.1462, -228.0, float("nan"), "FLO-MIN114"),
    ("pa_std_rna", "pa", (0.2, 0.9, 0.51, 0.53), (True, 79.2, 16.9), 0.1612, -251.0, 190.1, "FLO-PRO004RA"),
]


HEAD = 128   # the first f16 values of every output are kept beside the CRC-32 of all of them


def main():
    rs = reads()
    case_read, case_cfg, case_rna = [], [], []
    crc, head, n_out, f2, i2 = [], [], [], [], []
    for ri, (rname, x) in enumerate(rs):
        for ci, (cname, strat, q, std, scaling, offset, opl, fc) in enumerate(CONFIGS):
            for rna in (False, True):
                if rname.startswith("rna") != rna and cname in ("med_mad", "pa_raw") and not rname.endswith("peak"):
                    continue    # keep the fixture small: the cross combinations only for the main strategies
                r = O.ref_scaler_node(x, strat, q, std, rna, scaling, offset, opl, fc)
                case_read.append(ri); case_cfg.append(ci); case_rna.append(int(rna))
                bits = np.ascontiguousarray(r["signal"].view(np.uint16))
                crc.append(zlib.crc32(bits.tobytes())); head.append(np.resize(bits[:HEAD], HEAD) if bits.size else np.zeros(HEAD, np.uint16))
                n_out.append(bits.size)
                f2.append((r["scale_pa"], r["shift_pa"])); i2.append((r["num_trimmed_samples"], r["rna_adapter_end_signal_pos"]))
    np.savez_compressed(os.path.join(HERE, "scaler_node.npz"),
                        raw=np.concatenate([x for _, x in rs]), raw_len=np.array([x.size for _, x in rs], np.int64),
                        read_names=np.array([n for n, _ in rs]), case_read=np.array(case_read, np.int32),
                        case_cfg=np.array(case_cfg, np.int32), case_rna=np.array(case_rna, np.int32),
                        out_crc32=np.array(crc, np.uint32), out_head=np.stack(head), out_len=np.array(n_out, np.int64),
                        scale_shift_pa=np.array(f2, np.float32), trimmed_rna_end=np.array(i2, np.int32))
    i2 = np.array(i2)
    print(len(case_read), "cases;", "num_trimmed values:", sorted(set(i2[:, 0].tolist())))


if __name__ == "__main__":
    main()
