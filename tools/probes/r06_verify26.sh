#!/bin/bash
# residual pointwise layers after the plain-residual unswitch (r011 = residual + relu_out; round-6 reference: 64->256 r000 51.1 / r011 64.1-65.4 us,
# 256->64 46.4 / 51.5, 128->512 45.7 / 51.1)
O=gpurun_out/v27; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 hipcc -O2 -std=c++17 tools/conv_bench.cpp -I include -L xmem2_amd/csrc -lxmem_hip -Wl,-rpath,$R/xmem2_amd/csrc -o tools/conv_bench 2>&1 | tail -2
for r in 0,0,0 0,1,1; do
  timeout 120 tools/conv_bench -n 40 -r $r "4 120 216 64 256 1" 3 "4 120 216 256 64 1" 3 "4 60 108 128 512 1" 3 "4 30 54 256 1024 1" 3 "1 120 216 64 256 1" 3 "1 60 108 128 512 1" 3 2>&1 | grep "^shape" >> $O/pointwise.txt
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py -q -x > $O/tests.out 2>&1; echo "tests rc=$?" > $O/summary.txt
tail -2 $O/tests.out >> $O/summary.txt
for i in 1 2; do echo -n "b32 fp32: " >> $O/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt; done
echo -n "b32 no-prefetch: " >> $O/summary.txt
timeout 300 python bench.py --scale-only --steps 200 --no-prefetch 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt
cat $O/pointwise.txt $O/summary.txt
