#!/bin/bash
# Round 5, GPU call 2: full -m gpu suite (new: full-length decoder fixtures, whole-read pipeline fixture, int8 + variable chunks,
# fused layer tail vs f64), SIMD-partner stagger A/B (wsgemm + x8), cpu_baseline in the exact SURVEY 8d configuration (hac).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1
tail -25 $O/gpu_tests.log
timeout 600 python tools/stagger_ab.py > $O/stagger_ab.log 2>&1
cat $O/stagger_ab.log
( time timeout 900 python bench.py --model hac --steps 2 --warmup 1 --also-sup 0 --through-host 0 --cpu-baseline-full ) 2>$O/cpu_full.err | tail -1 > $O/cpu_baseline_full_hac.json
python - <<PY
import json
try:
    d = json.load(open("$O/cpu_baseline_full_hac.json"))
    print("cpu_baseline full:", d.get("cpu_baseline"))
except Exception as ex:
    print("cpu baseline full FAILED", ex)
PY
tail -3 $O/cpu_full.err
