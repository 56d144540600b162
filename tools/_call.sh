mkdir -p gpurun_out/r4p
timeout 600 python tools/q8_cluster_debug.py 512 3 none 2>&1 | grep -E "^C=" > gpurun_out/r4p/tests.log
timeout 900 python -m pytest tests/test_gpu_lstm_q8.py "tests/test_gpu_baseline_parity.py::test_quantised_cluster_lstm_vs_reference" -q -m gpu -s 2>&1 | grep -E "passed|failed|rror|C=|assert|case|spin|time" | cut -c1-900 >> gpurun_out/r4p/tests.log
timeout 300 python tools/stage_times.py --model sup --batch 8192 --steps 2 --quant 1 > gpurun_out/r4p/stage_sup_q8.json 2>&1
cat gpurun_out/r4p/tests.log; tail -n1 gpurun_out/r4p/stage_sup_q8.json
