#!/bin/bash
# kernel table of the timed region of the PLAIN step() path (no prefetch_keys hints): where value_no_prefetch goes
O=gpurun_out/v22; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --no-prefetch --no-cpu-baseline --no-extra-modes --plain-steps 0 --keep-trace $PWD/$O > $O/bench_plain.json 2> $O/bench_plain.err
ls $O
f=$(ls $O/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then timeout 120 python tools/trace_table.py "$f" > $O/plain_per_frame.csv 2>> $O/stats.err; timeout 120 python tools/stream_timeline.py "$f" > $O/plain_timeline.txt 2>> $O/stats.err; rm -f "$f"; fi
head -30 $O/plain_per_frame.csv | cut -c1-150
tail -c 600 $O/bench_plain.json
