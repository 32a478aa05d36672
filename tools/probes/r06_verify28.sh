#!/bin/bash
# one-launch HiddenUpdater input (xmem_hidden_update_gather): tests + A/B
O=gpurun_out/v28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py tests/test_gpu_e2e.py -q -x > $O/tests.out 2>&1; echo "tests rc=$?" > $O/summary.txt
tail -2 $O/tests.out >> $O/summary.txt
for v in 1 0 1 0; do echo -n "b32 fp32 XMEM_GATHER_HIDDEN_INPUT=$v: " >> $O/summary.txt
  XMEM_GATHER_HIDDEN_INPUT=$v timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt; done
for v in 1 0; do echo -n "b32 no-prefetch XMEM_GATHER_HIDDEN_INPUT=$v: " >> $O/summary.txt
XMEM_GATHER_HIDDEN_INPUT=$v timeout 300 python bench.py --scale-only --steps 200 --no-prefetch 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt; done
cat $O/summary.txt
