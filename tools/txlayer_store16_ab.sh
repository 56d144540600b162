#!/bin/bash
# A/B of the fused layer tail's 16-byte norm stores (csrc/txlayer.hip, TL_NORM_STORE16) on a GPU box.
# Run from the repo root IN THE BUILD CONTAINER first with "build": it leaves two libraries in the tree
# (dorado_amd/libmibc.so = product, dorado_amd/libmibc_store16.so = variant); then on the box:
#     bash tools/txlayer_store16_ab.sh run      -> gpurun_out/store16_ab.log
# Round 3 ended with the variant passing tests/test_gpu_txlayer.py and the sup@v5 BASELINE-size parity test, untimed.
set -e
case "${1:-run}" in
build)
    cd dorado_amd/csrc
    F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable"
    /opt/rocm/bin/hipcc $F -DTL_NORM_STORE16=1 -c txlayer.hip -o txlayer.store16.o
    make -s
    OBJS=$(ls *.o | grep -v -e '\.dbg\.o' -e 'txlayer\.o' -e 'txlayer\.store16\.o')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmibc_store16.so $OBJS txlayer.store16.o -ldl
    ;;
run)
    mkdir -p gpurun_out
    {
        echo "== product (8-byte stores)"
        timeout 120 python3 tools/txlayer_time.py 1048576 3 0x4003 | grep -E "^mode|stamps"
        echo "== variant (16-byte stores, data registers held until vmcnt(0))"
        MIBC_LIB=dorado_amd/libmibc_store16.so timeout 120 python3 tools/txlayer_time.py 1048576 3 0x4003 | grep -E "^mode|stamps"
    } 2>&1 | tee gpurun_out/store16_ab.log
    ;;
esac
