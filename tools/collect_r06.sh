#!/bin/bash
# Round-6 profiles of bench.py on the MI355X box (run from the repo root: bash tools/collect_r06.sh [part]).
# Writes to gpurun_out/prof_r06/ ; the summaries to keep are then copied into profiles/ and committed.
set -u
R=r06
PART=${1:-all}
OUT=$PWD/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
if [ $PART = all ] || [ $PART = a ]; then
# 1. PMC passes first (HBM-side bytes per frame, MFMA busy of the timed region): the bench line below quotes them (digest-guarded)
#    (SKIP_PMC=1: the committed passes stand - they are digest-guarded, i.e. valid while the kernel sources are unchanged)
if [ -z "${SKIP_PMC:-}" ]; then
( cd /tmp && timeout 900 python $REPO/tools/pmc_bench.py --workload b32 --steps 30 --out $OUT/${R}_bench_b32_pmc_per_frame.json > $OUT/pmc.log 2>&1 )
cp $OUT/${R}_bench_b32_pmc_per_frame.json profiles/ 2>/dev/null
fi
# 2. the bench line itself (B32, default flags: traced child + CPU baseline + the two opt-in modes as labelled extra keys)
timeout 900 python bench.py --keep-trace $OUT > $OUT/${R}_bench_b32.json 2> $OUT/bench_b32.err
python tools/trace_table.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_per_frame.csv 2>> $OUT/stats.err
python tools/trace_cut.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_kernel_trace.csv 2>> $OUT/stats.err
# 3. rocprofv3 --kernel-trace --stats of the same command (whole process: preload, captures, warm-up, timed region, instrumented passes)
( cd /tmp && rm -rf /tmp/prof_stats && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-trace --plain-steps 0 --no-extra-modes > /dev/null 2> $OUT/stats.err )
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_b32_kernel_stats.csv 2>/dev/null
rocminfo 2>/dev/null | grep -m3 -i "marketing name\|gfx" > $OUT/${R}_agent_info.txt
fi
if [ $PART = all ] || [ $PART = b ]; then
# 4. the opt-in modes, one line each with its own traced timed region
timeout 600 python bench.py --precision fp16 --keep-trace $OUT/fp16 --cpu-frames 6 > $OUT/${R}_bench_b32_fp16_loop.json 2> $OUT/bench_fp16.err
python tools/trace_table.py $OUT/fp16/b32_kernel_trace.csv > $OUT/${R}_bench_b32_fp16_loop_timed_region_per_frame.csv 2>> $OUT/stats.err
timeout 600 python bench.py --precision fp32x --keep-trace $OUT/split --cpu-frames 6 > $OUT/${R}_bench_b32_split.json 2> $OUT/bench_split.err
python tools/trace_table.py $OUT/split/b32_kernel_trace.csv > $OUT/${R}_bench_b32_split_timed_region_per_frame.csv 2>> $OUT/stats.err
for wl in c3 c4; do
  timeout 900 python bench.py --workload $wl --precision fp16 --steps 100 --no-kernel-trace --cpu-frames 6 > $OUT/${R}_bench_${wl}_fp16_loop.json 2> $OUT/bench_${wl}_fp16.err
done
fi
if [ $PART = all ] || [ $PART = c ]; then
# 5. the other workloads of SURVEY 8(d) (oracle parity leg where the oracle fits - with its own thread-noise floor for several
#    objects -, a sampled readout check at C4 / C5) and the realistic-motion variant
for wl in b32dyn c3 b32motion; do
  timeout 900 python bench.py --workload $wl --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 600 python bench.py --workload c4 --steps 100 --trace-steps 30 > $OUT/${R}_bench_c4.json 2> $OUT/bench_c4.err
timeout 900 python bench.py --workload c5 --steps 40 --no-kernel-trace --plain-steps 0 > $OUT/${R}_bench_c5.json 2> $OUT/bench_c5.err
fi
if [ $PART = all ] || [ $PART = d ]; then
# 6. isolated layer timings through the C ABI (no torch): classic F(4x4) tile vs the streaming GEMM, the direct form for scale
S1="1 120 216 256 256"; S2="1 60 108 512 512"; S3="1 60 108 512 256"; S4="1 30 54 576 512"; S5="4 120 216 64 64"; S6="4 60 108 128 128"; S7="4 30 54 256 256"; S8="1 30 54 512 512"; S9="4 30 54 1024 512"; S10="1 30 54 1600 512"
for sh in "$S1" "$S2" "$S3" "$S4" "$S5" "$S6" "$S7" "$S8" "$S9" "$S10"; do
  timeout 120 tools/conv_bench -n 30 "$sh" 3,9,19,23,26,24 >> $OUT/${R}_conv_bench_isolated_layers.txt 2>&1
done
fi
if [ $PART = all ] || [ $PART = e ]; then
# 7. run_on_video on files (decode + H2D + step + PNG masks; the reference harness's entry point, unchanged signature)
timeout 600 python tools/video_e2e_probe.py 400 > $OUT/${R}_run_on_video_files.txt 2>&1
fi
ls -la $OUT
