mkdir -p gpurun_out/r4f
( time timeout 400 python -m pytest tests -m gpu -q ) > gpurun_out/r4f/gpu_tests.log 2>&1
tail -6 gpurun_out/r4f/gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r4f/gpu_tests.log
STEPS="stats" SPECS="sup5:0:1024:12288" timeout 200 bash tools/refresh_profiles.sh r04_e
STEPS="bench" timeout 420 bash tools/refresh_profiles.sh r04_e
