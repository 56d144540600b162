#!/bin/bash
# Round 5, GPU call 11: tx_layer_kernel<3, 8> = token rows (attn, x in; x out) with the non-temporal cache policy, against <3, 0>.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_k
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3 4; do timeout 200 python tools/txlayer_time.py 1048576 3 0x803 2>&1 | grep "^mode"; done | tee $O/txlayer_nt_ab.log
