#!/bin/bash
mkdir -p gpurun_out/v9
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$R/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
{
for r in 0,0,0 0,1,1 0,0,1; do
tools/conv_bench -n 40 -ref 3 -r $r "4 120 216 64 256 1" 3,6,2,35,38 "4 120 216 256 64 1" 3,6,35 "4 120 216 64 64 1" 3 "4 60 108 128 512 1" 3,35 "4 60 108 512 128 1" 3,35 "4 30 54 1024 256 1" 3,6,35 "4 30 54 256 1024 1" 3,35 2>&1 | grep "^shape"
done
tools/conv_bench -n 30 -r 1,0,0 "1 30 54 512 512" 23 "1 120 216 256 256" 23 2>&1 | grep "^shape"
tools/conv_bench -n 30 -r 0,1,1 "1 30 54 512 512" 23,32,9,14 "1 60 108 256 256" 23 "1 120 216 256 256" 23 2>&1 | grep "^shape"
} > gpurun_out/v9/layers.txt
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py -q -x > gpurun_out/v9/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v9/summary.txt
tail -3 gpurun_out/v9/tests.out >> gpurun_out/v9/summary.txt
for rep in 1 2; do echo -n "b32: " >> gpurun_out/v9/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v9/summary.txt; done
cat gpurun_out/v9/layers.txt gpurun_out/v9/summary.txt
