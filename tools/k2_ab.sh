set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_s
(time timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder") > gpurun_out/r05_s/decoder_tests.log 2>&1; tail -4 gpurun_out/r05_s/decoder_tests.log
for i in 1 2; do
  for m in hac:16384 sup:8192 sup5:1024; do
    for v in 1 2; do
      MIBC_K2_V=$v timeout 120 python tools/stage_times.py --lib dbg --model ${m%%:*} --batch ${m##*:} --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${m%%:*} MIBC_K2_V=$v (1 = 32-lane kernel, 2 = 64-lane kernel) decode', d['decode'], 'total', d['total'])"
    done
  done
done 2>&1 | tee gpurun_out/r05_s/k2_ab.log
