mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py -q -m gpu -k "decod or baseline or beam or exact" 2>&1 | tail -2 > gpurun_out/r4w/t.log
for i in 1 2; do timeout 300 python tools/stage_times.py --model hac --batch 16384 --steps 3 2>&1 | tail -1 | cut -c1-100 >> gpurun_out/r4w/t.log; done
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4w/st -o s -- python $R/tools/stage_times.py --model hac --batch 16384 --steps 3 > /dev/null 2>&1
f=$(find $R/gpurun_out/r4w/st -name '*kernel_stats.csv' | head -1); grep -E "beam|bwd_scan|posts" $f | cut -c1-120 >> $R/gpurun_out/r4w/t.log; rm -rf $R/gpurun_out/r4w/st
cat $R/gpurun_out/r4w/t.log
