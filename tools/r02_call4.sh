#!/bin/bash
# cluster LSTM: correctness + timing + selected ablations
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_cluster_lstm.py -q -x > $O/cluster.log 2>&1
echo "cluster rc=$?"; tail -4 $O/cluster.log
for d in ${ABL:-0 1 2 16 27}; do
  echo -n "dbg=$d " ; MIBC_CL_DBG=$d timeout 200 python tools/stage_times.py --lib dbg --model sup --batch 8192 --steps 2 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['lstm_layer'], d['total'])"
done | tee $O/ablate.txt
