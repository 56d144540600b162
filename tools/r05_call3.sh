#!/bin/bash
# Round 5, GPU call 3: full -m gpu suite on the final code, the f64 layer-tail numbers, then everything profiles/ holds for the
# five configurations (bench line, rocprofv3 --kernel-trace --stats, PMC FETCH_SIZE / WRITE_SIZE passes) as r05_c.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1
tail -8 $O/gpu_tests.log
timeout 300 python -m pytest tests/test_gpu_txlayer.py -q -s -k f64 2>&1 | grep -E "fused layer tail|passed|failed" | tee $O/txlayer_f64.log
STEPS="bench stats pmc" timeout 1500 bash tools/refresh_profiles.sh r05_c 2>&1 | tail -60
