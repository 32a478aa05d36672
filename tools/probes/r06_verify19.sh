#!/bin/bash
O=gpurun_out/v19; mkdir -p $O
timeout 600 python tools/probes/filter_sizes.py b32 c4 c5 > $O/filter_sizes.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_affinity_served_sizes.py tests/test_gpu_ops.py tests/test_gpu_memory.py -q -x > $O/tests_a.out 2>&1; echo "tests_a rc=$?" > $O/summary.txt
tail -3 $O/tests_a.out >> $O/summary.txt
cat $O/filter_sizes.txt $O/summary.txt
