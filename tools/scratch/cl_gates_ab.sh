#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for q in 1 0; do
    echo -n "old q$q "; python tools/stage_times.py --model sup --quant $q --batch 8192 --steps 3 --lib dorado_amd/libmibc_ab_old.so 2>&1 | tail -1 | cut -c1-150
    echo -n "new q$q "; python tools/stage_times.py --model sup --quant $q --batch 8192 --steps 3 2>&1 | tail -1 | cut -c1-150
  done
done
