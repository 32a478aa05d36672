#!/bin/bash
mkdir -p gpurun_out/early3
run() { tag=$1; shift; echo "=== $tag: $*" >> gpurun_out/early3/summary.txt
  ( env "$@" timeout 400 python tools/probes/r06_early_probe2.py > gpurun_out/early3/$tag.out 2> gpurun_out/early3/$tag.err ); echo "rc=$?" >> gpurun_out/early3/summary.txt
  grep -h "parity\] GPU stream again\|PROBE" gpurun_out/early3/$tag.out gpurun_out/early3/$tag.err | cut -c1-260 >> gpurun_out/early3/summary.txt; }
run stub_sleep XMEM_EARLY_READOUT=1
run stub_nosleep XMEM_EARLY_READOUT=1 PROBE_DELAY=0
run stub_burn XMEM_EARLY_READOUT=1 PROBE_BURN=1
run real XMEM_EARLY_READOUT=1 PROBE_STUB=0
cat gpurun_out/early3/summary.txt
