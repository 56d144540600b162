cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -m gpu -x -q -k "decod or e2e or identity or beam" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e -o e -- python $R/tools/stage_times.py --steps 2 > $R/gpurun_out/prof_e.log 2>&1
tail -2 $R/gpurun_out/prof_e.log
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_e/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
