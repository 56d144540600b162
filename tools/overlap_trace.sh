#!/bin/bash
# One GPU-box call: kernel trace of bench.py with the decoder on its own stream (mibc_set_decode_overlap), reduced to a timeline
# (tools/overlap_timeline.py).  usage on the box: bash tools/overlap_trace.sh <tag>
set -u
TAG=${1:-r05_r}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in ${MODELS:-hac sup5}; do
  timeout 170 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$M -o t -- \
      python $R/bench.py --model $M --steps 3 --warmup 1 --also-sup 0 --through-host 0 --no-cpu-baseline --profile-run --decode-overlap 1 \
      > $O/trace_$M.log 2>&1
  echo "rc $? ($M)"; tail -1 $O/trace_$M.log | cut -c1-300
  f=$(find $O/trace_$M -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && head -2 "$f" | cut -c1-400
  python $R/tools/overlap_timeline.py $O/trace_$M $O/${TAG}_decode_overlap_timeline_$M.txt | head -40
  rm -rf $O/trace_$M
done
