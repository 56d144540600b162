#!/bin/bash
mkdir -p gpurun_out/r02k
timeout 600 python -m pytest tests/test_gpu_ws_lstm.py -q -m gpu -s -x 2>&1 | tail -15 > gpurun_out/r02k/test.log
cat gpurun_out/r02k/test.log
for d in 0; do
  MIBC_WS_LSTM_DBG=$d timeout 300 python tools/stage_times.py --lib dbg --steps 2 2>&1 | tail -1 >> gpurun_out/r02k/times.log
done
MIBC_WS_MIN_ROWS=100000000 timeout 300 python tools/stage_times.py --lib dbg --steps 2 2>&1 | tail -1 >> gpurun_out/r02k/times.log
cat gpurun_out/r02k/times.log
