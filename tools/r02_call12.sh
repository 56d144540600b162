#!/bin/bash
mkdir -p gpurun_out/r02k
rm -f gpurun_out/r02k/times.log
for d in 0 64; do
  MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=$d timeout 300 python tools/stage_times.py --lib dbg --steps 1 2>&1 | tail -2 | cut -c1-330 | grep -v amdgpu >> gpurun_out/r02k/times.log
done
cat gpurun_out/r02k/times.log
