#!/bin/bash
# cluster LSTM v4: var-chunk (masked) tests, BASELINE parity, full sup bench leg + rocprof stats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_variable_chunks.py tests/test_gpu_cluster_lstm.py -q -m gpu > $O/var.log 2>&1
echo "var+cluster rc=$?"; tail -4 $O/var.log
timeout 600 python -m pytest tests/test_gpu_baseline_parity.py -q > $O/parity.log 2>&1
echo "parity rc=$?"; tail -3 $O/parity.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sup -o s -- \
    python $R/bench.py --model sup --steps 2 --warmup 1 --no-cpu-baseline --also-sup 0 > $O/prof_sup.log 2>&1
echo "prof sup rc=$?"
f=$(find $O/prof_sup -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/r02e_kernel_stats_sup.csv && head -8 $O/r02e_kernel_stats_sup.csv
tail -1 $O/prof_sup.log | cut -c1-900
rm -rf $O/prof_sup
