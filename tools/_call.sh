mkdir -p gpurun_out/r4y
timeout 900 python -m pytest tests/test_gpu_lstm_q8.py "tests/test_gpu_baseline_parity.py::test_quantised_cluster_lstm_vs_reference" "tests/test_gpu_baseline_parity.py::test_quantised_lstm_vs_reference" -q -m gpu -s 2>&1 | grep -E "passed|failed|rror|C=|assert|case" | cut -c1-700 > gpurun_out/r4y/t.log
timeout 300 python tools/stage_times.py --model hac --batch 16384 --steps 2 --quant 1 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4y/t.log
timeout 300 python tools/stage_times.py --model sup --batch 8192 --steps 2 --quant 1 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4y/t.log
cat gpurun_out/r4y/t.log
