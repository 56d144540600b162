#!/bin/bash
# same-box sweep: SIMD-partner stagger of the int8 LSTM kernel (debug library, MIBC_Q8_STAGGER x 64 cycles)
cd $GRAFT_REPO_ROOT
for sg in 0 16 32 48 64 96 0 48; do
  echo -n "stagger $sg "; MIBC_Q8_STAGGER=$sg python tools/stage_times.py --model hac --quant 1 --batch 16384 --steps 3 --lib dbg 2>&1 | tail -1 | cut -c1-200
done
