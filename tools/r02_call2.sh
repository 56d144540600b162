#!/bin/bash
# Round-2 GPU call 2: cluster LSTM kernel correctness + first timing.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_cluster_lstm.py -q -s -x > $O/cluster.log 2>&1
echo "cluster rc=$?"; tail -15 $O/cluster.log
timeout 600 python -m pytest tests/test_variable_chunks.py -q -m gpu > $O/var.log 2>&1
echo "var rc=$?"; tail -8 $O/var.log
timeout 300 python tools/stage_times.py --model sup --batch 8192 --steps 2 > $O/stage_sup_cl.json 2> $O/stage_sup_cl.err
echo "stage rc=$?"; cat $O/stage_sup_cl.json; tail -3 $O/stage_sup_cl.err
timeout 600 python -m pytest tests/test_gpu_baseline_parity.py -q > $O/parity.log 2>&1
echo "parity rc=$?"; tail -5 $O/parity.log
