#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_gemm256.py -q -m gpu 2>&1 | tail -2
python tools/gemm_bench.py 0 2>&1 | grep -v amdgpu
python tools/gemm_ablate.py 2>&1 | grep -v amdgpu | head -3
timeout 600 python tools/stage_times.py --model sup5 --batch 1024 --steps 2 2>&1 | tail -1 | cut -c1-250
timeout 600 python tools/stage_times.py --model sup --batch 8192 --steps 1 2>&1 | tail -1 | cut -c1-250
