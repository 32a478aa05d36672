#!/bin/bash
mkdir -p gpurun_out/early2
run() { tag=$1; shift; echo "=== $tag: $*" >> gpurun_out/early2/summary.txt
  ( env "$@" timeout 300 python tools/probes/r06_early_probe.py > gpurun_out/early2/$tag.out 2> gpurun_out/early2/$tag.err ); echo "rc=$?" >> gpurun_out/early2/summary.txt
  grep -h PROBE gpurun_out/early2/$tag.out | cut -c1-300 >> gpurun_out/early2/summary.txt; tail -2 gpurun_out/early2/$tag.err | cut -c1-300 >> gpurun_out/early2/summary.txt; }
run on_delay XMEM_EARLY_READOUT=1 PROBE_DUMP=1
run on_nodelay XMEM_EARLY_READOUT=1 PROBE_DELAY=0 PROBE_DUMP=1
run off_delay XMEM_EARLY_READOUT=0 PROBE_DUMP=1
run on_delay_emptycache XMEM_EARLY_READOUT=1 PROBE_EMPTY_CACHE=1
run on_delay_nohint XMEM_EARLY_READOUT=1 XMEM_AFFINITY_HINT=0
run on_delay_noplain XMEM_EARLY_READOUT=1 PROBE_PLAIN=0
run on_delay_skip XMEM_EARLY_READOUT=1 XMEM_BENCH_SKIP_PASSES=1
cat gpurun_out/early2/summary.txt
