#!/bin/bash
for k in 1 0 1 0; do
  MIBC_DECODE_STAGGER=$k timeout 600 python tools/stage_times.py --lib dbg --model hac --batch 16384 --steps 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['decode'], d['head'], d['total'])"
done
for k in 1 0; do
  MIBC_DECODE_STAGGER=$k timeout 600 python tools/stage_times.py --lib dbg --model sup --batch 8192 --steps 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['decode'], d['head'], d['total'])"
done
