#!/bin/bash
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temp" | head -8
for k in 1 2; do
python bench.py --no-cpu-baseline --also-sup 0 --through-host 0 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_last_step']['lstm_layer'][:5], d['stage_ms_last_step']['decode'])"
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
