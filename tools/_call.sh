mkdir -p gpurun_out/r4ab
timeout 600 python tools/attention_ablate.py 256 > gpurun_out/r4ab/att.txt 2>&1
cat gpurun_out/r4ab/att.txt
