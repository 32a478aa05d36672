#!/bin/bash
mkdir -p gpurun_out/v2
timeout 1500 python -m pytest tests/test_gpu_c5_stream.py tests/test_gpu_host_ranks.py -x -q -s > gpurun_out/v2/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v2/summary.txt
grep -v "^$" gpurun_out/v2/tests.out | tail -25 | cut -c1-400 >> gpurun_out/v2/summary.txt
for rep in 1 2; do
for v in "XMEM_EARLY_READOUT=0" "XMEM_EARLY_READOUT=1" "XMEM_EARLY_READOUT=1 XMEM_BENCH_SAFE_HINTS=1" "XMEM_EARLY_READOUT=0 XMEM_BENCH_SAFE_HINTS=1"; do
  echo -n "b32 $v: " >> gpurun_out/v2/summary.txt
  env $v timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v2/summary.txt
done; done
cat gpurun_out/v2/summary.txt
