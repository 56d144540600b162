mkdir -p gpurun_out/r4t && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python $R/bench.py ) > $R/gpurun_out/r4t/bench.json 2> $R/gpurun_out/r4t/bench.err
tail -5 $R/gpurun_out/r4t/bench.err; python - <<'PY'
import json,os
l=[x for x in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4t/bench.json') if x.startswith('{')]
d=json.loads(l[-1])
print("value %.4g ms/step %.1f"%(d['value'], d['ms_per_step']), d['stage_ms_last_step'])
print("roofline", {k:d['roofline'][k] for k in ('kernel','frac','launch_ms','traffic')})
print("cpu", d.get('cpu_baseline'))
for k in ('through_host','through_host_multi_chunk_reads','through_host_variable'):
    v=d.get(k,{}); print(k, {a:v[a] for a in v if a!='what'})
for k,v in d.get('extra',{}).items():
    if 'error' in v: print(k, v); continue
    print(k, "%.4g"%v['samples_per_s'], "ms %.1f"%v['ms_per_step'], v['stage_ms_last_step'], v['roofline']['frac'], v['parity'].get('ok'), (v.get('cpu_baseline') or {}).get('value'))
PY
