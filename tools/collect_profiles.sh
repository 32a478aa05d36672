#!/bin/bash
# Round profiles of bench.py on the MI355X box (run from the repo root: bash tools/collect_profiles.sh r02).
# Writes to gpurun_out/prof_<round>/ ; the summaries to keep are then copied into profiles/ and committed.
set -u
R=${1:-r03}
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
# 4. the other workloads of SURVEY 8(d) (with the oracle parity leg where the oracle fits; a sampled readout check at C4 / C5),
#    the realistic-motion variant, and the two opt-in arithmetic modes (one bench line each)
for wl in b32dyn c3 b32motion; do
  timeout 900 python bench.py --workload $wl --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 600 python bench.py --workload c4 --steps 100 --trace-steps 30 > $OUT/${R}_bench_c4.json 2> $OUT/bench_c4.err
timeout 900 python bench.py --workload c5 --steps 40 --no-kernel-trace --plain-steps 0 > $OUT/${R}_bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python bench.py --precision fp32x --keep-trace $OUT/split > $OUT/${R}_bench_b32_split.json 2> $OUT/bench_split.err
python tools/trace_table.py $OUT/split/b32_kernel_trace.csv > $OUT/${R}_bench_b32_split_timed_region_per_frame.csv 2>> $OUT/stats.err
( cd /tmp && timeout 900 python $REPO/tools/pmc_bench.py --workload b32 --precision fp32x --steps 30 --out $OUT/${R}_bench_b32_split_pmc_per_frame.json > $OUT/pmc_split.log 2>&1 )
timeout 600 python bench.py --precision fp16 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_b32_fp16_mode.json 2> $OUT/bench_fp16.err
# 5. the split-operand mode layer by layer against float64, and the augmented preload
timeout 600 python tools/split_layer_errors.py > $OUT/${R}_split_conv_errors.txt 2>> $OUT/stats.err
timeout 600 python tools/preload_probe.py > $OUT/${R}_preload_augmented_480p.txt 2>> $OUT/stats.err
rocminfo 2>/dev/null | grep -m3 -i "marketing name\|gfx" > $OUT/${R}_agent_info.txt
ls -la $OUT
