#!/bin/bash
# final round-4 collection at the frozen kernel digest: parts a (B32 line, trace, stats, PMC), c (other workloads), d (isolated layers),
# and the fp16-loop lines
cd $GRAFT_REPO_ROOT
bash tools/collect_r04.sh a > gpurun_out/collect_a.log 2>&1
bash tools/collect_r04.sh d > gpurun_out/collect_d.log 2>&1
bash tools/collect_r04.sh c > gpurun_out/collect_c.log 2>&1
R=r04; OUT=gpurun_out/prof_r04
timeout 600 python bench.py --precision fp16 --keep-trace $OUT/fp16 --cpu-frames 6 > $OUT/${R}_bench_b32_fp16_loop.json 2> $OUT/bench_fp16.err
python tools/trace_table.py $OUT/fp16/b32_kernel_trace.csv > $OUT/${R}_bench_b32_fp16_loop_timed_region_per_frame.csv 2>> $OUT/stats.err
for wl in c3 c4; do
  timeout 900 python bench.py --workload $wl --precision fp16 --steps 100 --no-kernel-trace --cpu-frames 6 > $OUT/${R}_bench_${wl}_fp16_loop.json 2> $OUT/bench_${wl}_fp16.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/prof_r04/r04_bench_*.json")):
    if 'pmc' in f: continue
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); p=j.get("parity") or {}
        print(f.split('/')[-1], round(j["value"],1), j.get("value_no_prefetch"), j.get("value_fp32x"), j.get("value_fp16_loop"), {k:p.get(k) for k in ("mask_iou_vs_cpu_min","argmax_mismatch_pixels")}, j["roofline"].get("frac"), j["roofline"].get("frac_median"), j["roofline"].get("traffic"))
    except Exception as e: print(f, 'ERR', e)
PY
