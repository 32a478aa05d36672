#!/bin/bash
R=r06; OUT=$PWD/gpurun_out/prof_$R; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --keep-trace $OUT > $OUT/${R}_bench_b32.json 2> $OUT/bench_b32.err
python tools/trace_table.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_per_frame.csv 2>> $OUT/stats.err
python tools/trace_cut.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_kernel_trace.csv 2>> $OUT/stats.err
rm -f $OUT/b32_kernel_trace.csv
hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$PWD/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
bash tools/collect_r06.sh d > /dev/null 2>&1
bash tools/collect_r06.sh e > /dev/null 2>&1
tail -c 400 $OUT/${R}_bench_b32.json; tail -5 $OUT/${R}_run_on_video_files.txt
