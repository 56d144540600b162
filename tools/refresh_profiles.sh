#!/bin/bash
# One GPU-box call that regenerates what profiles/ holds for the GPU configurations of BASELINE.json (+ the opt-in int8 LSTM runs).
# usage (on the GPU box): bash tools/refresh_profiles.sh <tag>      e.g. r06_j          (STEPS="bench stats pmc" selects parts)
#   <tag>_bench.json                         the default bench.py line (hac headline + extra.*)
#   <tag>_kernel_stats_<model>_n<N>.csv      rocprofv3 --kernel-trace --stats of bench.py --model <model> [--quant 1] --profile-run
#   <tag>_pmc_traffic_<model>_n<N>.json      HBM bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only)
set -u
TAG=${1:-r06_x}
STEPS=${STEPS:-"bench stats pmc"}
SPECS=${SPECS:-"hac:0:16384:9996 sup:0:8192:9996 sup5:0:1024:12288 hac:1:16384:9996 sup:1:8192:9996"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
case " $STEPS " in *" bench "*)
  timeout 1500 python $R/bench.py 2>$O/bench.err | tail -1 > $O/${TAG}_bench.json
  cut -c1-400 $O/${TAG}_bench.json;;
esac
for spec in $SPECS; do
  IFS=: read M Q N TIN <<< "$spec"
  MK=$M; [ "$Q" = 1 ] && MK=${M}_q8
  case " $STEPS " in *" stats "*)
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$MK -o s -- \
        python $R/bench.py --model $M --quant $Q --steps 3 --warmup 1 --also-sup 0 --through-host 0 --no-cpu-baseline --profile-run > $O/stats_$MK.log 2>&1
    f=$(find $O/stats_$MK -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_${MK}_n$N.csv && head -5 $O/${TAG}_kernel_stats_${MK}_n$N.csv | cut -c1-140
    rm -rf $O/stats_$MK;;
  esac
  case " $STEPS " in *" pmc "*)
    timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$MK -o p -- \
        python $R/tools/stage_times.py --model $M --quant $Q --batch $N --steps 1 > $O/fetch_$MK.log 2>&1
    timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$MK -o p -- \
        python $R/tools/stage_times.py --model $M --quant $Q --batch $N --steps 1 > $O/write_$MK.log 2>&1
    python $R/tools/pmc_traffic.py $O/fetch_$MK $O/write_$MK $O/${TAG}_pmc_traffic_${MK}_n$N.json $MK $N $TIN | head -12
    rm -rf $O/fetch_$MK $O/write_$MK;;
  esac
done
