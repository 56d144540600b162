#!/bin/bash
# Full GPU validation on a box: the -m gpu suite and smoke().
mkdir -p gpurun_out/suite
(time timeout 600 python -m pytest tests -q -m gpu) 2>&1 | tail -8 | tee gpurun_out/suite/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/suite/smoke.log
