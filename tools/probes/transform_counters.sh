#!/bin/bash
# HBM-side bytes and durations of the F(4x4) transform kernels and the streaming position GEMM on two layers (tools/conv_bench, plan 23):
# separate --pmc passes for FETCH_SIZE / WRITE_SIZE, one --kernel-trace --stats pass for the durations.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "1 120 216 256 256" "4 120 216 64 64"; do
  echo "== conv_bench \"$shape\" plan 23"
  for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pm
    timeout 60 rocprofv3 --pmc $pass --output-format csv -d /tmp/pm -- $R/tools/conv_bench -n 20 "$shape" 23 > /tmp/pm.log 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    for k in wino4_input_kernel wino4_output_kernel gemm_stream_kernel; do
      [ -n "$f" ] && python3 $R/tools/pmc_kernel.py $f $k | sed "s/^/$k  /"
    done
  done
  rm -rf /tmp/pm
  timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- $R/tools/conv_bench -n 20 "$shape" 23 > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/kstats.py $f | grep -i "wino4\|gemm_stream"
done
