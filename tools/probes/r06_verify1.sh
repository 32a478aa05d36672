#!/bin/bash
mkdir -p gpurun_out/v1
( XMEM_EARLY_READOUT=1 timeout 400 python tools/probes/r06_early_probe2.py > gpurun_out/v1/stub.out 2> gpurun_out/v1/stub.err ); echo "stub rc=$?" > gpurun_out/v1/summary.txt
grep -h "parity\] GPU stream again\|PROBE" gpurun_out/v1/stub.out gpurun_out/v1/stub.err | cut -c1-200 >> gpurun_out/v1/summary.txt
timeout 900 python -m pytest tests/test_gpu_prefetch_sync.py tests/test_gpu_stream_b32.py -x -q -s > gpurun_out/v1/tests.out 2>&1; echo "tests rc=$?" >> gpurun_out/v1/summary.txt
tail -15 gpurun_out/v1/tests.out >> gpurun_out/v1/summary.txt
timeout 1500 python tests/parity_by_plan.py 480p_3obj_bench_c3 > gpurun_out/v1/c3_parity_by_plan.txt 2>&1; echo "pbp rc=$?" >> gpurun_out/v1/summary.txt
cat gpurun_out/v1/c3_parity_by_plan.txt >> gpurun_out/v1/summary.txt
cat gpurun_out/v1/summary.txt
