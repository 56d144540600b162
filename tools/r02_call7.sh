#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "two_phase or host_layer" > $O/host.log 2>&1
echo "host tests rc=$?"; tail -4 $O/host.log
for m in hac sup sup5; do timeout 400 python tools/through_host_bench.py --model $m 2>&1 | tail -1; done | tee $O/through_host.txt
