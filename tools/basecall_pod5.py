#!/usr/bin/env python3
"""BASELINE configs[0] plumbing: POD5 file(s) -> device VBZ decode -> ScalerNode parameters (host formula) ->
raw int16 chunks scaled inside conv1 -> LSTM-CRF / transformer -> beam search -> stitched reads (FASTQ on stdout).
Weights are random-init (no model files offline), so the bases are meaningless; the point is that every
stage of the reference's front end (DataLoader -> ScalerNode -> BasecallerNode) has a device-side
counterpart here.  usage: basecall_pod5.py <pod5 file or dir> [--model hac|fast|tiny] [--batch 64]"""
import argparse
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import capi, config, hostapi, pod5, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--model", default="hac")
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    files = [a.path] if os.path.isfile(a.path) else sorted(glob.glob(os.path.join(a.path, "**", "*.pod5"), recursive=True))
    cfg = {"hac": config.hac_v43, "fast": config.fast_v43, "tiny": lambda: config.tiny(128, 4)}[a.model]()
    ws = synth.make_weights(cfg, seed=1)
    eng = capi.Engine(cfg, ws)
    t0 = time.perf_counter()
    reads = []
    for p in files:
        f = pod5.Pod5File(p)
        reads += f.load_signals(f.reads(), eng)
    eng.close()
    t1 = time.perf_counter()
    raws, ss, ts = [], [], []
    for r in reads:
        sc = hostapi.pa_read_scaling(True, 93.69, 23.5, r.scaling, r.offset, r.open_pore_level, r.flow_cell_product_code)
        ss.append((sc["shift"] + sc["open_pore_adjustment"], sc["scale"]))
        ts.append(10 if r.raw.size > 10 else 0)
        raws.append(r.raw)
    called, stats = hostapi.basecall_raw_reads(cfg, ws, raws, np.array(ss, np.float32), ts, batch_size=a.batch)
    t2 = time.perf_counter()
    for r, c in zip(reads, called):
        print(f"@{r.read_id} ch={r.channel} mux={r.mux} ns={r.num_samples} sr={r.sample_rate}\n{c[0]}\n+\n{c[1]}")
    n = sum(r.num_samples for r in reads)
    print(f"# {len(reads)} reads, {n} samples; load+decode {1e3 * (t1 - t0):.1f} ms, basecall {1e3 * (t2 - t1):.1f} ms "
          f"(includes engine creation); stats {stats}", file=sys.stderr)


if __name__ == "__main__":
    main()
