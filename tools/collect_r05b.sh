#!/bin/bash
# Round-5 lines after the early readout became opt-in (default off): the B32 line with its trace tables, and the workloads whose parity leg matters.
set -u
R=r05; OUT=$PWD/gpurun_out/prof_r05b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_selector.py tests/test_gpu_stream_b32.py -q -x > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 600 python bench.py --keep-trace $OUT > $OUT/${R}_bench_b32.json 2> $OUT/bench_b32.err
python tools/trace_table.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_per_frame.csv 2>> $OUT/stats.err
python tools/trace_cut.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_kernel_trace.csv 2>> $OUT/stats.err
timeout 400 python bench.py --workload c3 --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_c3.json 2> $OUT/bench_c3.err
timeout 300 python bench.py --workload b32dyn --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_b32dyn.json 2> $OUT/bench_b32dyn.err
timeout 300 python bench.py --workload c4 --steps 100 --trace-steps 30 > $OUT/${R}_bench_c4.json 2> $OUT/bench_c4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r05b/r05_bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); p=j.get('parity') or {}
        print(f.split('/')[-1], round(j['value'],1), j.get('value_no_prefetch'), p.get('mask_iou_vs_cpu_min'), p.get('argmax_mismatch_pixels'), j['config'].get('early_readout'))
    except Exception as e: print(f, 'ERR', e)
PY
