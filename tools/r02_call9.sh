#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_adapter.py tests/test_host_cpu.py -q > $O/adapter.log 2>&1
echo "adapter rc=$?"; tail -5 $O/adapter.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1
echo "gpu suite rc=$?"; tail -6 $O/gpu_all.log
