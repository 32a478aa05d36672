#!/bin/bash
mkdir -p gpurun_out/early4
run() { tag=$1; shift; echo "=== $tag: $*" >> gpurun_out/early4/summary.txt
  ( env "$@" timeout 400 python tools/probes/r06_early_probe3.py > gpurun_out/early4/$tag.out 2> gpurun_out/early4/$tag.err ); echo "rc=$?" >> gpurun_out/early4/summary.txt
  grep -h "parity\] GPU stream again\|PROBE" gpurun_out/early4/$tag.out gpurun_out/early4/$tag.err | cut -c1-200 >> gpurun_out/early4/summary.txt; tail -3 gpurun_out/early4/$tag.err | cut -c1-300 >> gpurun_out/early4/summary.txt; }
run dump XMEM_EARLY_READOUT=1
cat gpurun_out/early4/summary.txt
