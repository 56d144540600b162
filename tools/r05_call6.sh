#!/bin/bash
# Round 5, GPU call 6 (final code): full -m gpu suite, then everything profiles/ holds for the five configurations as r05_f.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_f
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
STEPS="bench stats pmc" timeout 1500 bash tools/refresh_profiles.sh r05_f 2>&1 | tail -12
