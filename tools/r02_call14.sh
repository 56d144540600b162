#!/bin/bash
timeout 1800 python -m pytest tests/test_gpu_gemm256.py tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_attention.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/stage_times.py --model sup5 --batch 1024 --steps 2 2>&1 | tail -1 | cut -c1-250
