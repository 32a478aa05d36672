#!/bin/bash
mkdir -p gpurun_out/c15
for d in 0 1 5; do
  echo "== XMEM_F16_PIPE=1 XMEM_F16_DBG=$d" >> gpurun_out/c15/knockouts.txt
  PROBE_NOCHECK=$d XMEM_F16_DBG=$d timeout 200 python tools/probes/filter_sizes.py b32 c4 c5 2>&1 | grep -v amdgpu.ids >> gpurun_out/c15/knockouts.txt
done
cat gpurun_out/c15/knockouts.txt
