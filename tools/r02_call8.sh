#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
for n in 1024 2048 4096; do timeout 300 python tools/stage_times.py --model sup5 --batch $n --steps 2 | cut -c1-200; done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02h/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","parity")}); print(d["roofline"]); print(d.get("cpu_baseline")); print(d.get("through_host"))
for k,v in d.get("extra",{}).items():
    print(k, {kk:v.get(kk) for kk in ("samples_per_s","ms_per_step","parity","cpu_baseline","error")}); print(v.get("roofline"))
PY
