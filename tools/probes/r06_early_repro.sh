#!/bin/bash
# Round 6: reproduce / classify the early-readout wrong-mask stream of bench.py's parity leg.
# Each variant is its own process on the same box; only the [parity] lines and the value are kept.
mkdir -p gpurun_out/early
run() {
  tag=$1; shift
  echo "=== $tag: $*" >> gpurun_out/early/summary.txt
  ( env "$@" XMEM_BENCH_PARITY_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-modes --cpu-frames 4 $EXTRA \
      > gpurun_out/early/$tag.out 2> gpurun_out/early/$tag.err )
  echo "rc=$?" >> gpurun_out/early/summary.txt
  grep -h "\[parity\]" gpurun_out/early/$tag.err | cut -c1-400 >> gpurun_out/early/summary.txt
  python - <<PY >> gpurun_out/early/summary.txt
import json
try:
    l=[x for x in open('gpurun_out/early/$tag.out') if x.startswith('{')][-1]
    j=json.loads(l); print('value', j['value'], 'early', j['config']['early_readout'], 'iou_min', j['parity']['mask_iou_vs_cpu_min'], 'mismatch', j['parity']['argmax_mismatch_pixels'])
except Exception as e:
    print('no line', e)
PY
}
EXTRA=""
run base_on XMEM_EARLY_READOUT=1
EXTRA="--no-kernel-trace"
run notrace_on XMEM_EARLY_READOUT=1
run serialize_on XMEM_EARLY_READOUT=1 AMD_SERIALIZE_KERNEL=3
run eagerseg_on XMEM_EARLY_READOUT=1 XMEM_EAGER_STAGES=segment
run off XMEM_EARLY_READOUT=0
cat gpurun_out/early/summary.txt
