#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$R/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
timeout 900 bash tools/collect_r06.sh d > /dev/null 2>&1
timeout 900 bash tools/collect_r06.sh e > /dev/null 2>&1
tail -5 gpurun_out/prof_r06/r06_run_on_video_files.txt; grep -c "^shape" gpurun_out/prof_r06/r06_conv_bench_isolated_layers.txt
