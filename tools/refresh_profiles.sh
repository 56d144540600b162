#!/bin/bash
# One GPU-box call that regenerates what profiles/ holds for the three GPU configurations of BASELINE.json.
# usage (on the GPU box): bash tools/refresh_profiles.sh <tag>      e.g. r02_h
#   <tag>_bench.json                         the default bench.py line (hac headline + extra.sup_v43 / sup_v50)
#   <tag>_kernel_stats_<model>_n<N>.csv      rocprofv3 --kernel-trace --stats of bench.py --model <model>
#   <tag>_pmc_traffic_<model>_n<N>.json      HBM bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes)
set -u
TAG=${1:-r03_x}
PMC_MODELS=${PMC_MODELS:-"hac sup sup5"}      # PMC passes only for these (unchanged kernels keep their earlier profile)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py 2>$O/bench.err | tail -1 > $O/${TAG}_bench.json
cut -c1-600 $O/${TAG}_bench.json
for spec in "hac 16384 9996" "sup 8192 9996" "sup5 1024 12288"; do
  set -- $spec; M=$1; N=$2; TIN=$3
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$M -o s -- \
      python $R/bench.py --model $M --steps 3 --warmup 1 --also-sup 0 --through-host 0 --no-cpu-baseline --profile-run > $O/stats_$M.log 2>&1
  f=$(find $O/stats_$M -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_${M}_n$N.csv && head -6 $O/${TAG}_kernel_stats_${M}_n$N.csv
  case " $PMC_MODELS " in *" $M "*) ;; *) rm -rf $O/stats_$M; continue;; esac
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$M -o p -- \
      python $R/tools/stage_times.py --model $M --batch $N --steps 1 > $O/fetch_$M.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$M -o p -- \
      python $R/tools/stage_times.py --model $M --batch $N --steps 1 > $O/write_$M.log 2>&1
  python $R/tools/pmc_traffic.py $O/fetch_$M $O/write_$M $O/${TAG}_pmc_traffic_${M}_n$N.json $M $N $TIN
  rm -rf $O/stats_$M $O/fetch_$M $O/write_$M
done
