#!/bin/bash
O=gpurun_out/v36; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --workload c5 --steps 20 --trace-steps 8 --plain-steps 0 --no-cpu-baseline --no-extra-modes --keep-trace $PWD/$O > $O/bench_c5.json 2> $O/bench_c5.err
f=$(ls $O/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then timeout 120 python tools/trace_table.py "$f" > $O/c5_per_frame.csv 2>> $O/stats.err; rm -f "$f"; fi
head -45 $O/c5_per_frame.csv | cut -c1-170; tail -3 $O/bench_c5.err
