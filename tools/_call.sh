mkdir -p gpurun_out/r4a
tools/mfma_tile_clock.bin 200000 > gpurun_out/r4a/mfma_tile_clock.jsonl 2>&1
bash tools/txlayer_store16_ab.sh run > /dev/null 2>&1; cp gpurun_out/store16_ab.log gpurun_out/r4a/
for m in sup sup5 hac; do b=16384; [ $m = sup ] && b=8192; [ $m = sup5 ] && b=1024; timeout 300 python tools/stage_times.py --model $m --batch $b --steps 2 > gpurun_out/r4a/stage_$m.json 2>&1; done
cat gpurun_out/r4a/mfma_tile_clock.jsonl gpurun_out/r4a/store16_ab.log; tail -n1 gpurun_out/r4a/stage_*.json
