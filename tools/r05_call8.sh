#!/bin/bash
# Round 5, GPU call 8: hac LSTM, non-temporal x_t loads + h_t stores (MIBC_LSTM_DBG=56) against the plain copy (8), five alternations.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python tools/nt_ab.py hac 2>&1 | tee $O/nt_ab_x8.log
