#!/bin/bash
# knock-outs of the pipelined filter kernel (library built with -DXMEM_TOOLS)
mkdir -p gpurun_out/c14
for d in 0 1 2 3 4 5 7; do
  echo "== XMEM_F16_PIPE=1 XMEM_F16_DBG=$d" >> gpurun_out/c14/knockouts.txt
  PROBE_NOCHECK=1 XMEM_F16_DBG=$d timeout 200 python tools/probes/filter_sizes.py b32 c4 2>&1 | grep -v amdgpu.ids >> gpurun_out/c14/knockouts.txt
done
cat gpurun_out/c14/knockouts.txt
