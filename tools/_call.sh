mkdir -p gpurun_out/r4z
timeout 900 python -m pytest tests/test_gpu_lstm_q8.py tests/test_gpu_multi.py -q -m gpu -s 2>&1 | grep -E "passed|failed|rror|C=|assert" | cut -c1-300 > gpurun_out/r4z/t.log
cat gpurun_out/r4z/t.log
