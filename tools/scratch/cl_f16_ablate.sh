#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 1 96 0; do
  echo -n "f16 MIBC_CL_DBG=$d "; MIBC_CL_DBG=$d python tools/stage_times.py --model sup --quant 0 --batch 8192 --steps 2 --lib dbg 2>&1 | tail -1 | cut -c1-160
done
