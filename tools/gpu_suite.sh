#!/bin/bash
# Full GPU validation on a box: the -m gpu suite, the experimental-kernel tests (opt-in), smoke().
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/suite/gpu_tests.log
MIBC_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_ws_lstm.py -q -m gpu 2>&1 | tail -2 | tee gpurun_out/suite/experimental.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/suite/smoke.log
