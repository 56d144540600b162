#!/bin/bash
# Round 5, GPU call 10 (final code: non-temporal activation streams in the x8 and q8 LSTM kernels): full -m gpu suite, then everything
# profiles/ holds for the five configurations as r05_j.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_j
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
STEPS="bench stats pmc" timeout 1500 bash tools/refresh_profiles.sh r05_j 2>&1 | tail -12
