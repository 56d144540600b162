#!/bin/bash
mkdir -p gpurun_out/r02j
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "batch_size" -s 2>&1 | tail -8 > gpurun_out/r02j/test.log
timeout 500 python tools/auto_batch.py > gpurun_out/r02j/auto_batch.jsonl 2> gpurun_out/r02j/auto_batch.err
cat gpurun_out/r02j/test.log gpurun_out/r02j/auto_batch.jsonl; tail -3 gpurun_out/r02j/auto_batch.err
