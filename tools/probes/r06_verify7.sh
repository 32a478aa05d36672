#!/bin/bash
mkdir -p gpurun_out/v7
timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_gpu_e2e.py tests/test_gpu_stream_b32.py -q -x > gpurun_out/v7/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v7/summary.txt
tail -4 gpurun_out/v7/tests.out >> gpurun_out/v7/summary.txt
for rep in 1 2 3; do echo -n "b32: " >> gpurun_out/v7/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v7/summary.txt; done
echo -n "b32 steps 20: " >> gpurun_out/v7/summary.txt
timeout 300 python bench.py --scale-only --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v7/summary.txt
cat gpurun_out/v7/summary.txt
