#!/bin/bash
mkdir -p gpurun_out/v15
timeout 2000 python -m pytest tests/test_gpu_prefetch_sync.py tests/test_gpu_e2e.py tests/test_gpu_stream_b32.py tests/test_gpu_memory.py tests/test_gpu_harness.py -q -x > gpurun_out/v15/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v15/summary.txt
tail -5 gpurun_out/v15/tests.out >> gpurun_out/v15/summary.txt
for v in "XMEM_PLAIN_OVERLAP=1" "XMEM_PLAIN_OVERLAP=0" "XMEM_PLAIN_OVERLAP=1" "XMEM_PLAIN_OVERLAP=0"; do echo -n "b32 no-prefetch $v: " >> gpurun_out/v15/summary.txt
  env $v timeout 300 python bench.py --scale-only --steps 200 --no-prefetch 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v15/summary.txt; done
cat gpurun_out/v15/summary.txt
