#!/bin/bash
mkdir -p gpurun_out/v6
tools/probes/fused_intensity/feed_sweep > gpurun_out/v6/feed_sweep.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_e2e.py -q -x -k "not noise_floor and not plain_checkpoint and not consolidation_vs_oracle" > gpurun_out/v6/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v6/summary.txt
tail -4 gpurun_out/v6/tests.out >> gpurun_out/v6/summary.txt
for rep in 1 2; do for v in "XMEM_FUSE_HIDDEN_UPDATE=1" "XMEM_FUSE_HIDDEN_UPDATE=0"; do echo -n "b32 $v: " >> gpurun_out/v6/summary.txt
  env $v timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v6/summary.txt; done; done
cat gpurun_out/v6/feed_sweep.txt gpurun_out/v6/summary.txt
