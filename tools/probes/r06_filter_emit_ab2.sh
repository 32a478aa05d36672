#!/bin/bash
# the list-emitting filter on longer lists (hint from perturbed queries; a block of queries with a garbage hint): knock-outs in a tools build
O=gpurun_out/v24; mkdir -p $O
XMEM_HIPCC_FLAGS=-DXMEM_TOOLS timeout 600 python -m xmem2_amd.build --force > $O/build_tools.log 2>&1
for nz in 0.05 0.15 0.3; do for hv in "" "512:640"; do for d in 0 8 16 32; do
  echo "== noise $nz heavy '$hv' XMEM_F16_DBG=$d" >> $O/filter_ab2.txt
  PROBE_HEAVY=$hv PROBE_NOISE=$nz XMEM_F16_DBG=$d PROBE_NOCHECK=1 timeout 120 python tools/probes/filter_sizes.py b32 2>&1 | grep "lists\|^b32" >> $O/filter_ab2.txt
done; done; done
cat $O/filter_ab2.txt
