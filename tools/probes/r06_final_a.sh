#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "conv2d" > gpurun_out/final_conv_tests.out 2>&1; tail -2 gpurun_out/final_conv_tests.out
timeout 2000 bash tools/collect_r06.sh a > /dev/null 2>&1; ls gpurun_out/prof_r06 | head -20
