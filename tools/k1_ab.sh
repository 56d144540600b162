set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_q
(time timeout 900 python -m pytest tests -x -q -m gpu) > gpurun_out/r05_q/gpu_tests.log 2>&1; tail -4 gpurun_out/r05_q/gpu_tests.log
for i in 1 2 3; do
  for m in hac:16384 sup5:1024; do
    timeout 300 python tools/stage_times.py --lib dbg --model ${m%%:*} --batch ${m##*:} --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${m%%:*} depth-first LSE (debug library) decode', d['decode'], 'total', d['total'])"
    timeout 300 python tools/stage_times.py --model ${m%%:*} --batch ${m##*:} --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${m%%:*} breadth-first LSE (product)      decode', d['decode'], 'total', d['total'])"
  done
done 2>&1 | tee gpurun_out/r05_q/k1_lse_ab.log
