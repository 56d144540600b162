#!/bin/bash
# cluster LSTM ablations (debug library): MIBC_CL_DBG bits 1 no gates, 2 no DMA, 4 no MFMA, 8 no hand-off, 16 no frag reads
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
for d in 0 1 8 9 2 4 16 18 26 27; do
  echo -n "dbg=$d " ; MIBC_CL_DBG=$d timeout 200 python tools/stage_times.py --lib dbg --model sup --batch 8192 --steps 2 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['lstm_layer'], d['total'])"
done | tee $O/ablate.txt
