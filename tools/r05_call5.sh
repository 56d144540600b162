#!/bin/bash
# Round 5, GPU call 5: conv2 on the f32-input MFMA — bit identity against the VALU loop, conv stage time A/B (product library vs
# debug library with MIBC_CONV12_VALU=1) on hac and sup, then the full -m gpu suite.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv12.py -q 2>&1 | tail -15 | tee $O/conv12_identity.log
for i in 1 2; do
  for m in hac sup; do
    b=16384; [ $m = sup ] && b=8192
    echo "== $m MFMA conv2 (product)"; timeout 300 python tools/stage_times.py --model $m --batch $b --steps 3 2>&1 | tail -1
    echo "== $m VALU conv2 (debug library, MIBC_CONV12_VALU=1)"; MIBC_CONV12_VALU=1 timeout 300 python tools/stage_times.py --lib dbg --model $m --batch $b --steps 3 2>&1 | tail -1
    echo "== $m MFMA conv2 (debug library)"; timeout 300 python tools/stage_times.py --lib dbg --model $m --batch $b --steps 3 2>&1 | tail -1
  done
done 2>&1 | tee $O/conv12_ab.log
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1
tail -8 $O/gpu_tests.log
