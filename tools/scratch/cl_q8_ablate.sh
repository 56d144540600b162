#!/bin/bash
# int8 cluster LSTM timing ablations (debug library; results wrong): 1 no gate math, 8 no hand-off wait, 32 / 64 / 96 weight / activation /
# both slabs always fetched from one cache-resident address
cd $GRAFT_REPO_ROOT
for d in 0 96 32 64 8 1 0; do
  echo -n "MIBC_CL_DBG=$d "; MIBC_CL_DBG=$d python tools/stage_times.py --model sup --quant 1 --batch 8192 --steps 2 --lib dbg 2>&1 | tail -1 | cut -c1-160
done
