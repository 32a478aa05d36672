#!/bin/bash
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_affinity_served_sizes.py tests/test_gpu_rows16.py tests/test_gpu_stream_b32.py -q -x -m gpu > gpurun_out/c23/pytest.log 2>&1
tail -3 gpurun_out/c23/pytest.log
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "affinity or readout or memory" >> gpurun_out/c23/pytest.log 2>&1
tail -3 gpurun_out/c23/pytest.log
timeout 200 python tools/probes/filter_sizes.py b32 c4 c5 2>&1 | grep -v amdgpu.ids > gpurun_out/c23/filter_sizes.txt; cat gpurun_out/c23/filter_sizes.txt
