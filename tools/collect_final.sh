set -u
R=r03
OUT=$PWD/gpurun_out/prof_${R}_final
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 900 python $REPO/tools/pmc_bench.py --workload b32 --steps 30 --out $OUT/${R}_bench_b32_pmc_per_frame.json > $OUT/pmc.log 2>&1 )
( cd /tmp && timeout 900 python $REPO/tools/pmc_bench.py --workload b32 --precision fp32x --steps 30 --out $OUT/${R}_bench_b32_split_pmc_per_frame.json > $OUT/pmc_split.log 2>&1 )
cp $OUT/${R}_bench_b32_pmc_per_frame.json $OUT/${R}_bench_b32_split_pmc_per_frame.json profiles/
timeout 600 python bench.py --keep-trace $OUT > $OUT/${R}_bench_b32.json 2> $OUT/bench_b32.err
python tools/trace_table.py $OUT/b32_kernel_trace.csv > $OUT/${R}_bench_b32_timed_region_per_frame.csv 2>> $OUT/stats.err
( cd /tmp && rm -rf /tmp/prof_stats && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-trace --plain-steps 0 > /dev/null 2> $OUT/stats.err )
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_b32_kernel_stats.csv 2>/dev/null
timeout 600 python bench.py --precision fp32x --keep-trace $OUT/split > $OUT/${R}_bench_b32_split.json 2> $OUT/bench_split.err
python tools/trace_table.py $OUT/split/b32_kernel_trace.csv > $OUT/${R}_bench_b32_split_timed_region_per_frame.csv 2>> $OUT/stats.err
rm -f $OUT/b32_kernel_trace.csv $OUT/split/b32_kernel_trace.csv
# the large-memory workloads (their readout is dominated by the filter kernel)
timeout 600 python bench.py --workload c4 --steps 100 --trace-steps 30 > $OUT/${R}_bench_c4.json 2> $OUT/bench_c4.err
timeout 900 python bench.py --workload c5 --steps 40 --no-kernel-trace --plain-steps 0 > $OUT/${R}_bench_c5.json 2> $OUT/bench_c5.err
for wl in c3 b32motion; do
  timeout 900 python bench.py --workload $wl --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_$wl.json 2> $OUT/bench_$wl.err
done
ls -la $OUT
