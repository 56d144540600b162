mkdir -p gpurun_out/r4q
for r in 1 2; do
for lib in "" dorado_amd/libmibc_wrow0.so; do for q in 0 1; do
echo "lib=${lib:-default(wrow1)} quant=$q" >> gpurun_out/r4q/ab.txt
timeout 300 python tools/stage_times.py --model sup --batch 8192 --steps 2 --quant $q ${lib:+--lib $lib} 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4q/ab.txt
done; done; done
timeout 600 python -m pytest tests/test_gpu_cluster_lstm.py tests/test_gpu_lstm_q8.py -q -m gpu 2>&1 | tail -2 >> gpurun_out/r4q/ab.txt
cat gpurun_out/r4q/ab.txt
