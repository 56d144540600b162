#!/bin/bash
# Round 5, GPU call 1: full -m gpu suite on the event-ring + decode-overlap code, decode-overlap A/B (three configs),
# x8 LSTM variants (MIBC_X8_VAR), tx_layer_kernel<2,4> (every fourth weight fragment from L2).  Every step has its own timeout.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
timeout 900 python tools/x8_var_check.py --steps 3 > $O/x8_var.log 2>&1
cat $O/x8_var.log
for m in hac sup sup5; do
  for ov in 0 1 0 1; do
    st=8; [ $m = sup ] && st=4
    timeout 300 python bench.py --model $m --steps $st --warmup 2 --also-sup 0 --through-host 0 --no-cpu-baseline --decode-overlap $ov 2>$O/ovl_${m}_$ov.err | tail -1 > $O/ovl_${m}_$ov.json
    python - <<PY
import json
try:
    d = json.load(open("$O/ovl_${m}_$ov.json"))
    print("$m overlap=$ov", round(d["ms_per_step"], 2), "ms/step", "%.4g" % d["value"], "parity", d["parity"].get("ok"), d["stage_ms_last_step"]["lstm_layer"][:5])
except Exception as ex:
    print("$m overlap=$ov FAILED", ex)
PY
  done
done 2>&1 | tee $O/ovl_summary.log
timeout 600 python tools/txlayer_time.py 1048576 2 0x402 0x4402 > $O/txlayer_l2frag.log 2>&1
cat $O/txlayer_l2frag.log
