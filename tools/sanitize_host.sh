#!/bin/bash
# Host-layer logic under sanitizers, no GPU:  bash tools/sanitize_host.sh [thread|address]   (default: both)
# Builds tools/host_node_sanitize.cpp + dorado_amd/host/*.cpp with -fsanitize=... against the prebuilt libmibc.so (never called)
# into /tmp and runs it; exits non-zero on any report.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/mibc_sanitize
mkdir -p $O
for kind in ${1:-thread address}; do
  flags="-fsanitize=thread -fPIE -pie"
  [ "$kind" = address ] && flags="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
  g++ -std=c++17 -O1 -g $flags -I$R/dorado_amd/host -I$R/include $R/tools/host_node_sanitize.cpp $R/dorado_amd/host/mibc_host.cpp \
      $R/dorado_amd/host/tensor_loader.cpp -L$R/dorado_amd -lmibc -Wl,-rpath,$R/dorado_amd -lpthread -o $O/host_$kind
  TSAN_OPTIONS="halt_on_error=1 exitcode=66" ASAN_OPTIONS="detect_leaks=0" $O/host_$kind
  echo "sanitize_host: $kind clean"
done
