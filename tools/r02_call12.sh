#!/bin/bash
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_ws_lstm.py -q -m gpu 2>&1 | tail -1; done
