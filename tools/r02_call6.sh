#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_gemm256.py -q -s > $O/gemm256.log 2>&1
echo "gemm256 rc=$?"; grep "differing\|passed\|failed\|Error" $O/gemm256.log | head -20
timeout 300 python tools/gemm_bench.py 0 2>&1 | tee $O/gemm_bench.txt
timeout 600 python -m pytest tests/test_gpu_baseline_parity.py -q -k "sup" > $O/parity.log 2>&1
echo "parity rc=$?"; tail -3 $O/parity.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "transformer" > $O/tx.log 2>&1
echo "tx rc=$?"; tail -3 $O/tx.log
for m in sup sup5; do timeout 300 python tools/stage_times.py --model $m --batch $([ $m = sup ] && echo 8192 || echo 1024) --steps 2; done
