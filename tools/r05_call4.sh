#!/bin/bash
# Round 5, GPU call 4: the one test whose tolerance changed, then everything profiles/ holds as r05_d (bench knee fixed for the
# quantised wide layers: call 3 ran the sup@v4.3 int8 extra at 22528 chunks instead of 8192).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_variable_chunks.py -m gpu -q 2>&1 | tail -4 | tee $O/var_tests.log
STEPS="bench stats pmc" timeout 1500 bash tools/refresh_profiles.sh r05_d 2>&1 | tail -70
