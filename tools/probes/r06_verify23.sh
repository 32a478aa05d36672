#!/bin/bash
O=gpurun_out/v23; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/probes/filter_sizes.py b32 c4 c5 > $O/filter_sizes.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_affinity_served_sizes.py tests/test_gpu_ops.py tests/test_gpu_memory.py -q -x > $O/tests_a.out 2>&1; echo "tests_a rc=$?" > $O/summary.txt
tail -3 $O/tests_a.out >> $O/summary.txt
timeout 600 python bench.py --no-prefetch --no-cpu-baseline --no-extra-modes --plain-steps 0 --keep-trace $PWD/$O > $O/bench_plain.json 2> $O/bench_plain.err
f=$(ls $O/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then timeout 120 python tools/trace_table.py "$f" > $O/plain_per_frame.csv 2>> $O/stats.err; rm -f "$f"; fi
grep "^b32\|^c4\|^c5" $O/filter_sizes.txt; cat $O/summary.txt; grep "affinity\|^# timed" $O/plain_per_frame.csv | cut -c1-150
