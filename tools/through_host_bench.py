#!/usr/bin/env python3
"""Samples/s through the C++ host layer (SimplexBasecaller -> HipModelRunner -> HipCaller -> mibc_call_async):
synthetic single-chunk reads in host memory, 2 runners, two batches in flight.  Compare with bench.py's
device-resident number.   python tools/through_host_bench.py [--model hac] [--batch 0] [--batches 6]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import config, hostapi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="hac")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--batches", type=int, default=6)
ap.add_argument("--runners", type=int, default=2)
a = ap.parse_args()
cfg = {"hac": config.hac_v43, "sup": config.sup_v43, "sup5": config.sup_v50}[a.model]()
ws = synth.make_weights(cfg, seed=42)
reads = synth.make_signal(256, cfg.chunk_size, seed=77)
nb = a.batch or {"hac": 16384, "sup": 8192, "sup5": 1024}[a.model]
r = hostapi.bench_through_host(cfg, ws, reads, n_warm=2 * nb, n_reads=a.batches * nb, num_runners=a.runners, batch_size=nb)
r.update(model=cfg.name, batch=nb, runners=a.runners)
print(json.dumps(r))
