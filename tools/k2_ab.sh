set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_n
(time timeout 900 python -m pytest tests -x -q -m gpu) > gpurun_out/r05_n/gpu_tests.log 2>&1; tail -4 gpurun_out/r05_n/gpu_tests.log
for i in 1 2 3; do for ht in 0 1; do for m in hac:16384 sup5:1024 sup:8192; do
  MIBC_K2_HT=$ht timeout 300 python tools/stage_times.py --lib dbg --model ${m%%:*} --batch ${m##*:} --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${m%%:*} k2_ht=$ht decode', d['decode'], 'total', d['total'])"
done; done; done 2>&1 | tee gpurun_out/r05_n/k2_ht_ab.log
