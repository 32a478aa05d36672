#!/bin/bash
mkdir -p gpurun_out/c19
O=gpurun_out/c19/filter_ab.txt
run() { echo "== $*" >> $O; env "$@" timeout 200 python tools/probes/filter_sizes.py c4 c5 2>&1 | grep -v amdgpu.ids >> $O; }
run XMEM_F16_PIPE=1 XMEM_F16_WAVES=4
run XMEM_F16_PIPE=1 XMEM_F16_WAVES=4 XMEM_F16_DBG=8 PROBE_NOCHECK=1
run XMEM_F16_PIPE=1 XMEM_F16_WAVES=4 XMEM_F16_DBG=13 PROBE_NOCHECK=1
cat $O
tools/probes/mfma_shadow/lds_feed 16000 2>&1 | head -9 | tail -3
