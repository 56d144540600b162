mkdir -p gpurun_out/r4g
timeout 18 python -m pytest tests/test_adapter.py -x -q -m gpu -k "basecaller_node and 1" -p no:cacheprovider > gpurun_out/r4g/node_var.log 2>&1
tail -15 gpurun_out/r4g/node_var.log
