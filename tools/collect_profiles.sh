#!/bin/bash
# Round profiles of bench.py on the MI355X box (run from the repo root: bash tools/collect_profiles.sh r02).
# Writes to gpurun_out/prof_<round>/ ; the summaries to keep are then copied into profiles/ and committed.
set -u
R=${1:-r02}
OUT=$PWD/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
# 1. the bench line itself (B32, default flags: kernel trace child + CPU baseline)
timeout 600 python bench.py --keep-trace $OUT > $OUT/${R}_bench_b32.json 2> $OUT/bench_b32.err
# 2. rocprofv3 --kernel-trace --stats of the same command (whole process: preload, captures, warm-up, timed region, instrumented pass)
( cd /tmp && rm -rf /tmp/prof_stats && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-trace > /dev/null 2> $OUT/stats.err )
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_b32_kernel_stats.csv 2>/dev/null
python tools/trace_table.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_per_frame.csv 2>> $OUT/stats.err
# 3. PMC passes (HBM-side bytes per frame, MFMA busy) of the timed region
( cd /tmp && timeout 900 python $REPO/tools/pmc_bench.py --workload b32 --steps 30 --out $OUT/${R}_bench_b32_pmc_per_frame.json > $OUT/pmc.log 2>&1 )
# 4. the other workloads of SURVEY 8(d) and the reduced-precision mode (one bench line each)
for wl in b32dyn c3 c4; do
  timeout 600 python bench.py --workload $wl --steps 100 --trace-steps 30 --no-cpu-baseline > $OUT/${R}_bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 900 python bench.py --workload c5 --steps 40 --no-kernel-trace --no-cpu-baseline > $OUT/${R}_bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python bench.py --precision fp16 --trace-steps 30 > $OUT/${R}_bench_b32_fp16_mode.json 2> $OUT/bench_fp16.err
rocminfo 2>/dev/null | grep -m3 -i "marketing name\|gfx" > $OUT/${R}_agent_info.txt
ls -la $OUT
