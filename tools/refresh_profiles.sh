#!/bin/bash
# One GPU-box call that regenerates everything profiles/ holds for the hac headline workload.
# usage (on the GPU box): bash tools/refresh_profiles.sh <tag>      e.g. r01_g
set -u
TAG=${1:-r01_x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. bench line (default flags)
python $R/bench.py 2>$O/bench.err | tail -1 > $O/${TAG}_bench_hac_sup_sup5.json
# 2. rocprofv3 kernel stats of the same command (hac leg only to bound the time)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --also-sup 0 --no-cpu-baseline > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_hac_n16384.csv
# 3. HBM traffic (separate PMC passes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python $R/tools/stage_times.py --steps 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- python $R/tools/stage_times.py --steps 1 > $O/write.log 2>&1
python $R/tools/pmc_traffic.py $O/fetch $O/write $O/${TAG}_pmc_traffic_hac_n16384.json hac 16384 9996
head -12 $O/${TAG}_kernel_stats_hac_n16384.csv
cat $O/${TAG}_bench_hac_sup_sup5.json | cut -c1-400
rm -rf $O/stats $O/fetch $O/write
