#!/bin/bash
# isolated per-kernel durations of the decoder's 3x3 layers (B32, one object) with their shipped plans: tools/conv_bench under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$R/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
cd /tmp
run() { # shape plan flags
  rm -rf /tmp/pm
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- $R/tools/conv_bench -n 30 -r $3 "$1" $2 > /tmp/pm.log 2>&1
  grep "^shape" /tmp/pm.log
  f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/kstats.py $f | grep -i "wino\|gemm_stream\|conv_mfma" | cut -c1-160
}
run "1 30 54 576 512" 23 1,0,0
run "1 30 54 512 512" 32 0,1,1
run "1 30 54 512 512" 23 1,0,0
run "1 60 108 512 256" 26 0,0,0
run "1 60 108 512 256" 23 0,1,1
run "1 60 108 256 256" 23 1,0,0
run "1 120 216 256 256" 23 1,0,0
run "1 120 216 256 256" 23 0,1,1
run "1 30 54 320 192" 22 0,0,0
