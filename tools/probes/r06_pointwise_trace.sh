#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp xmem2_amd/csrc/libxmem_hip.so /tmp/libxmem_hip.keep
XMEM_HIPCC_FLAGS=-DXMEM_TOOLS python -m xmem2_amd.build --force > /tmp/build_tools.log 2>&1 || { tail -5 /tmp/build_tools.log; exit 1; }
hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$R/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
for sh in "4 120 216 64 256 1" "4 120 216 256 64 1" "4 60 108 128 512 1" "4 60 108 512 128 1" "4 30 54 1024 256 1"; do
  for dbg in 8 9 11; do
    echo "== $sh  XMEM_CONV_DBG=$dbg"
    XMEM_CONV_DBG=$dbg tools/conv_bench -n 20 -ref 3 -r 0,0,0 "$sh" 3 2>&1 | grep "conv trace\|^shape" | cut -c1-330
  done
done
cp /tmp/libxmem_hip.keep xmem2_amd/csrc/libxmem_hip.so
