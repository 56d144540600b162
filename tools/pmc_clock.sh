#!/bin/bash
# One PMC pass per configuration: effective shader clock of every kernel (GRBM_GUI_ACTIVE / launch duration) and
# where its wave cycles go (SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY,
# matrix-pipe busy cycles, LDS-array cycles, bank conflicts).  Counters only (--kernel-trace for the durations): no --stats / sys-trace.
# usage (GPU box): bash tools/pmc_clock.sh <tag> ["hac:0:16384 hac:1:16384 sup:0:8192 sup:1:8192 sup5:0:1024"]
set -u
TAG=${1:-r06_x}
SPECS=${2:-"hac:0:16384 hac:1:16384 sup:0:8192 sup:1:8192 sup5:0:1024"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in $SPECS; do
  IFS=: read M Q N <<< "$spec"
  MK=$M; [ "$Q" = 1 ] && MK=${M}_q8
  timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --kernel-trace --output-format csv -d $O/clk_$MK -o p -- python $R/tools/stage_times.py --model $M --quant $Q --batch $N --steps 2 > $O/clk_$MK.log 2>&1
  python $R/tools/pmc_clock.py $O/clk_$MK $O/${TAG}_pmc_clock_${MK}_n$N.json $MK $N | head -14
  rm -rf $O/clk_$MK
done
