#!/bin/bash
# Round 5, GPU call 9: non-temporal A/B for the wsgemm activation loads and the int8 LSTM kernel (debug library switches).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_i
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python tools/nt_ab.py 2>&1 | tee $O/nt_ab2.log
