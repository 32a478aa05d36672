#!/bin/bash
mkdir -p gpurun_out/v13
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py tests/test_gpu_fp16_loop.py tests/test_gpu_e2e.py -q -x -k "not noise_floor and not plain_checkpoint and not consolidation_vs_oracle" > gpurun_out/v13/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v13/summary.txt
tail -3 gpurun_out/v13/tests.out >> gpurun_out/v13/summary.txt
git_head=none
for p in fp32 fp32 fp32; do echo -n "b32 $p: " >> gpurun_out/v13/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 --precision $p 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v13/summary.txt; done
timeout 600 python bench.py --no-cpu-baseline --no-extra-modes --keep-trace gpurun_out/v13 > gpurun_out/v13/line.json 2> gpurun_out/v13/err.txt; python tools/trace_table.py gpurun_out/v13/b32_kernel_trace.csv > gpurun_out/v13/per_frame.csv; rm -f gpurun_out/v13/b32_kernel_trace.csv
cat gpurun_out/v13/summary.txt; head -50 gpurun_out/v13/per_frame.csv
