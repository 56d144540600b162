#!/bin/bash
for k in 1 2; do MIBC_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_ws_lstm.py -q -m gpu 2>&1 | tail -1; done
for d in 0 64; do
  MIBC_WS_MIN_ROWS=2048 MIBC_WS_LSTM_DBG=$d timeout 300 python tools/stage_times.py --lib dbg --steps 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['lstm_layer'])"
done
MIBC_WS_MIN_ROWS=100000000 timeout 300 python tools/stage_times.py --lib dbg --steps 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x8', d['lstm_layer'])"
