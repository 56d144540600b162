#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<COUNTERS>" [env assignments...] ; runs stage_times --steps 1 under rocprofv3 --pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CNT=$2; shift 2
env "$@" rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o p -- python $R/tools/stage_times.py --steps 1 > $R/gpurun_out/pmc_$TAG.log 2>&1
python - "$R/gpurun_out/pmc_$TAG" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:48]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[k]+=1
for k,v in acc.items():
    print(k, 'launches', cnt[k], {a: f'{b/cnt[k]:.4g}' for a,b in v.items()})
PY
