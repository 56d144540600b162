#!/bin/bash
# Host-layer logic under sanitizers, no GPU:  bash tools/sanitize_host.sh [thread|address]   (default: both)
# Builds dorado_amd/host/*.cpp against the C-ABI test double tools/fake_mibc.cpp (never against libmibc.so) with
# -fsanitize=thread and with -fsanitize=address,undefined (+ leak check) into $TMPDIR and runs
#   tools/host_node_sanitize.cpp    the node over plain stand-in runners, scaler_node through its ops seam from 8 threads, row packer,
#                                   mutation fuzz of the .tensor parser on the reference's fixtures (tests/golden/tensor)
#   tools/host_caller_sanitize.cpp  the whole layer: HipCaller (GPU thread, device FIFO, two async slots), HipModelRunner (variable
#                                   packing, overflow batches), fixed / raw int16 / variable / two devices, scaler_node beside the node,
#                                   each cross-checked against a direct evaluation
# Exits non-zero on any sanitizer report or failed cross-check.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/mibc_sanitize
mkdir -p $O
SRC="$R/tools/fake_mibc.cpp $R/dorado_amd/host/mibc_host.cpp $R/dorado_amd/host/tensor_loader.cpp"
for kind in ${1:-thread address}; do
  flags="-fsanitize=thread -fPIE -pie"
  [ "$kind" = address ] && flags="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
  for drv in host_node_sanitize host_caller_sanitize; do
    g++ -std=c++17 -O1 -g $flags -I$R/dorado_amd/host -I$R/include $R/tools/$drv.cpp $SRC -lpthread -o $O/${drv}_$kind &
  done
  wait
  for drv in host_node_sanitize host_caller_sanitize; do
    TSAN_OPTIONS="halt_on_error=1 exitcode=66" ASAN_OPTIONS="detect_leaks=1" $O/${drv}_$kind $R
  done
  echo "sanitize_host: $kind clean"
done
