#!/bin/bash
mkdir -p gpurun_out/c22
O=gpurun_out/c22/filter_ab.txt
run() { echo "== $*" >> $O; env "$@" timeout 200 python tools/probes/filter_sizes.py b32 c4 c5 2>&1 | grep -v amdgpu.ids >> $O; }
run XMEM_F16_PIPE=1
run XMEM_F16_PIPE=0
run XMEM_F16_PIPE=1 XMEM_F16_WAVES=4
run XMEM_F16_PIPE=0 XMEM_F16_WAVES=4
run XMEM_F16_PIPE=1 XMEM_F16_DBG=8 PROBE_NOCHECK=1
run XMEM_F16_PIPE=1 XMEM_F16_DBG=5 PROBE_NOCHECK=1
run XMEM_F16_PIPE=1 XMEM_F16_DBG=13 PROBE_NOCHECK=1
cat $O
