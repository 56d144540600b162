"""Prints the timing-based batch-size sweep (HipCaller batch_size=-1, the reference's CudaCaller.cpp:552-627
procedure) for the three BASELINE configurations.  GPU only.  python tools/auto_batch.py [hac] [sup43] [sup5]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dorado_amd import config, hostapi, synth  # noqa: E402

CASES = {"hac": config.hac_v43, "sup43": config.sup_v43, "sup5": config.sup_v50}
for nm in sys.argv[1:] or list(CASES):
    cfg = CASES[nm]()
    ws = synth.make_weights(cfg, seed=42)
    knee, _ = hostapi.auto_batch_size(cfg, ws, mode=0)
    chosen, timings = hostapi.auto_batch_size(cfg, ws, mode=-1)
    print(json.dumps({"model": nm, "chunk_size": cfg.chunk_size, "formula_batch": knee, "timed_batch": chosen,
                      "sweep_ms_per_chunk": [[b, round(t, 6)] for b, t in timings]}), flush=True)
