#!/bin/bash
# Round 6: what bounds the pointwise layers of the batch-4 key encoder?  tools build (-DXMEM_TOOLS) with XMEM_CONV_DBG knock-outs
# (1 = no epilogue stores, 2 = every M-tile loads tile 0's A rows, 3 = both); wrong results, timing only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp xmem2_amd/csrc/libxmem_hip.so /tmp/libxmem_hip.keep
XMEM_HIPCC_FLAGS=-DXMEM_TOOLS python -m xmem2_amd.build --force > /tmp/build_tools.log 2>&1 || { tail -5 /tmp/build_tools.log; exit 1; }
hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$R/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
for dbg in 0 1 2 3; do
  echo "== XMEM_CONV_DBG=$dbg"
  XMEM_CONV_DBG=$dbg tools/conv_bench -n 40 -r 0,0,0 "4 120 216 64 256 1" 3,6 "4 120 216 256 64 1" 3 "4 60 108 128 512 1" 3 "4 60 108 512 128 1" 3 "4 30 54 1024 256 1" 3 2>&1 | grep "^shape"
  XMEM_CONV_DBG=$dbg tools/conv_bench -n 40 -r 0,1,1 "4 120 216 64 256 1" 3 "4 60 108 128 512 1" 3 2>&1 | grep "^shape"
done
cp /tmp/libxmem_hip.keep xmem2_amd/csrc/libxmem_hip.so
