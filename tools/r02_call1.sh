#!/bin/bash
# Round-2 GPU call 1 (on the GPU box): BASELINE-size parity, the whole GPU suite, the bench line, and
# baseline kernel statistics of the two sup configurations (before the round-2 kernel work).
#   gpurun --timeout 2400 -- 'bash tools/r02_call1.sh'
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py -q -s > $O/parity.log 2>&1
echo "parity rc=$?"; tail -5 $O/parity.log
timeout 1200 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_baseline_parity.py > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; cut -c1-1500 $O/bench.json
cd /tmp
for M in sup sup5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$M -o s -- \
      python $R/bench.py --model $M --steps 2 --warmup 1 --no-cpu-baseline --also-sup 0 > $O/prof_$M.log 2>&1
  echo "prof $M rc=$?"
  f=$(find $O/prof_$M -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/r02a_kernel_stats_$M.csv && head -14 $O/r02a_kernel_stats_$M.csv
  tail -1 $O/prof_$M.log | cut -c1-600
  rm -rf $O/prof_$M
done
