#!/bin/bash
mkdir -p gpurun_out/v3
timeout 1500 python -m pytest tests/test_gpu_c5_stream.py tests/test_gpu_host_ranks.py -q -s > gpurun_out/v3/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v3/summary.txt
grep -v "^$" gpurun_out/v3/tests.out | grep -v "^E   " | tail -12 | cut -c1-600 >> gpurun_out/v3/summary.txt
grep "AssertionError\|host ms\|C5 stream" gpurun_out/v3/tests.out | cut -c1-900 >> gpurun_out/v3/summary.txt
cat gpurun_out/v3/summary.txt
