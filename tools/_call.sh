mkdir -p gpurun_out/r4x
timeout 2000 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r4x/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r4x/gpu_tests.log
cat gpurun_out/r4x/gpu_tests.log
