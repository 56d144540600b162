mkdir -p gpurun_out/r4r
for r in 1 2; do
for lib in "" dorado_amd/libmibc_wsnt.so; do
echo "lib=${lib:-default}" >> gpurun_out/r4r/ab.txt
timeout 300 python tools/stage_times.py --model hac --batch 16384 --steps 3 ${lib:+--lib $lib} 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4r/ab.txt
done; done
cat gpurun_out/r4r/ab.txt
