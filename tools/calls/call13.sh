#!/bin/bash
# filter kernel A/B (two-waves-per-SIMD vs software-pipelined), then the tests that pin the readout + the fp16 loop
mkdir -p gpurun_out/c13
for p in 0 1; do
  echo "== XMEM_F16_PIPE=$p" >> gpurun_out/c13/filter_sizes.txt
  XMEM_F16_PIPE=$p timeout 300 python tools/probes/filter_sizes.py b32 c4 c5 >> gpurun_out/c13/filter_sizes.txt 2>&1
done
cat gpurun_out/c13/filter_sizes.txt
timeout 900 python -m pytest tests/test_gpu_affinity_served_sizes.py tests/test_gpu_rows16.py tests/test_gpu_fp16_loop.py -q -x -m gpu > gpurun_out/c13/pytest.log 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k affinity >> gpurun_out/c13/pytest.log 2>&1
tail -5 gpurun_out/c13/pytest.log
