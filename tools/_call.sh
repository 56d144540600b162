mkdir -p gpurun_out/r4aa
timeout 300 python tools/txlayer_time.py 1048576 2 0x102 0x202 3 2>&1 | grep -E "^mode|stamps" | cut -c1-200 > gpurun_out/r4aa/t.log
cat gpurun_out/r4aa/t.log
