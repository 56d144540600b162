#!/bin/bash
MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=128 timeout 300 python tools/ws_trace.py 2>&1 | grep -v amdgpu | tail -34
echo ---- cache-resident
MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=192 timeout 300 python tools/ws_trace.py 2>&1 | grep -v amdgpu | tail -12
