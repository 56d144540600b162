#!/bin/bash
MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=128 timeout 300 python tools/ws_trace.py 2>&1 | grep -v amdgpu | tail -4
for d in 0 64; do
  MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=$d timeout 300 python tools/stage_times.py --lib dbg --steps 1 2>&1 | tail -2 | cut -c1-330 | grep -v amdgpu
done
