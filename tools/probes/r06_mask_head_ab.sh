#!/bin/bash
O=gpurun_out/v35; mkdir -p $O; rm -f /tmp/mask_head_*.pt
XMEM_COUT1_ROW4=0 timeout 200 python tools/probes/mask_head_ab.py > $O/ab.txt 2>&1
XMEM_COUT1_ROW4=1 timeout 200 python tools/probes/mask_head_ab.py >> $O/ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py -q -x > $O/tests.out 2>&1; echo "tests rc=$?" >> $O/ab.txt; tail -2 $O/tests.out >> $O/ab.txt
for rep in 1 2 3; do for v in 1 0; do echo -n "b32 hinted XMEM_COUT1_ROW4=$v: " >> $O/ab.txt
  XMEM_COUT1_ROW4=$v timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/ab.txt; done; done
grep -v amdgpu.ids $O/ab.txt
