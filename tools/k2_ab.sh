set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_t
(time timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder or end_to_end or round_trip") > gpurun_out/r05_t/decoder_tests.log 2>&1; tail -4 gpurun_out/r05_t/decoder_tests.log
for i in 1 2; do
  for m in hac:16384 sup:8192 sup5:1024; do
    for lib in dorado_amd/libmibc_ab_k2v2.so dorado_amd/libmibc.so; do
      timeout 120 python tools/stage_times.py --lib $lib --model ${m%%:*} --batch ${m##*:} --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${m%%:*} $lib decode', d['decode'], 'total', d['total'])"
    done
  done
done 2>&1 | tee gpurun_out/r05_t/k2_ab2.log
