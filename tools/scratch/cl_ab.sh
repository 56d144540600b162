#!/bin/bash
# same-box A/B: round-5 library (libmibc_ab_r05.so: 32x32 MFMAs, (row >> 2) & 3 swizzle, separate int8 conversion pass, scalar gate
# math) vs the round-6 library, alternating; sup@v4.3 f16 / int8 and hac int8 / f16
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for spec in "sup 0 8192" "sup 1 8192" "hac 1 16384" "hac 0 16384"; do
    set -- $spec
    echo -n "r05 $1 q$2 "; python tools/stage_times.py --model $1 --quant $2 --batch $3 --steps 3 --lib dorado_amd/libmibc_ab_r05.so 2>&1 | tail -1 | cut -c1-170
    echo -n "r06 $1 q$2 "; python tools/stage_times.py --model $1 --quant $2 --batch $3 --steps 3 2>&1 | tail -1 | cut -c1-170
  done
done
