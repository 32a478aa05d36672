#!/bin/bash
# isolated durations of every kernel of the hinted select at B32 (rocprofv3 kernel trace of tools/probes/filter_sizes.py)
O=$PWD/gpurun_out/v20; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_sel && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sel -- python $GRAFT_REPO_ROOT/tools/probes/filter_sizes.py b32 > $O/probe.txt 2>&1 )
f=$(find /tmp/prof_sel -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -20 "$f" | cut -c1-160 > $O/select_kernel_stats.txt; else echo "no stats file" > $O/select_kernel_stats.txt; fi
grep "^b32" $O/probe.txt; cat $O/select_kernel_stats.txt
