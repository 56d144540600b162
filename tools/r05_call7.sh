#!/bin/bash
# Round 5, GPU call 7: non-temporal loads / stores A/B (debug library switches) for the decoder kernels and the hac LSTM.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/nt_ab.py hac 2>&1 | tee $O/nt_ab_hac.log
timeout 400 python tools/nt_ab.py sup5 2>&1 | tee $O/nt_ab_sup5.log
timeout 600 python tools/nt_ab.py sup 2>&1 | tee $O/nt_ab_sup.log
