#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r02m/gpu_tests.log
cat gpurun_out/r02m/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
