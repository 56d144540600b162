#!/bin/bash
# same-box A/B: hac int8 step with the round-5 gate math (libmibc_ab_old.so) vs packed / pre-scaled (libmibc.so), alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "old "; python tools/stage_times.py --model hac --quant 1 --batch 16384 --steps 3 --lib dorado_amd/libmibc_ab_old.so 2>&1 | tail -1
  echo -n "new "; python tools/stage_times.py --model hac --quant 1 --batch 16384 --steps 3 2>&1 | tail -1
done
