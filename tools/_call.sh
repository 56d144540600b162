mkdir -p gpurun_out/r4g
( time timeout 150 python -m pytest tests/test_scaler_node.py -x -q -m gpu ) > gpurun_out/r4g/scaler_gpu.log 2>&1
tail -12 gpurun_out/r4g/scaler_gpu.log
