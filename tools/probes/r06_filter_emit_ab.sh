#!/bin/bash
# round 6: where the time of the list-emitting filter goes (tools build: XMEM_F16_DBG 8 = no pushes, 16 = pushes but no drain, 32 = drain without atomics)
O=gpurun_out/v18; mkdir -p $O
echo "== shipped build" > $O/filter_ab.txt
timeout 600 python tools/probes/filter_sizes.py b32 c4 c5 >> $O/filter_ab.txt 2>&1
XMEM_HIPCC_FLAGS=-DXMEM_TOOLS python -m xmem2_amd.build --force > $O/build_tools.log 2>&1
for d in 0 8 16 32; do echo "== tools build, XMEM_F16_DBG=$d" >> $O/filter_ab.txt
  XMEM_F16_DBG=$d PROBE_NOCHECK=1 timeout 600 python tools/probes/filter_sizes.py b32 c4 >> $O/filter_ab.txt 2>&1; done
cat $O/filter_ab.txt
