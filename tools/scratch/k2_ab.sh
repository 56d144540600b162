#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "old "; python tools/stage_times.py --model hac --quant 1 --batch 16384 --steps 3 --lib dorado_amd/libmibc_ab_old.so 2>&1 | tail -1 | cut -c1-100
  echo -n "new "; python tools/stage_times.py --model hac --quant 1 --batch 16384 --steps 3 2>&1 | tail -1 | cut -c1-100
done
echo -n "old sup5 "; python tools/stage_times.py --model sup5 --batch 1024 --steps 3 --lib dorado_amd/libmibc_ab_old.so 2>&1 | tail -1 | cut -c1-100
echo -n "new sup5 "; python tools/stage_times.py --model sup5 --batch 1024 --steps 3 2>&1 | tail -1 | cut -c1-100
