mkdir -p gpurun_out/r4c
{ timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_lstm_q8.py "tests/test_gpu_baseline_parity.py" tests/test_gpu_parity.py -q -m gpu -s 2>&1 | grep -E "passed|failed|Error|error|N=|C=|assert|case" | cut -c1-600; } > gpurun_out/r4c/tests.log 2>&1
timeout 300 python tools/stage_times.py --model sup5 --batch 1024 --steps 2 > gpurun_out/r4c/stage_sup5.json 2>&1
cat gpurun_out/r4c/tests.log; tail -n1 gpurun_out/r4c/stage_sup5.json
