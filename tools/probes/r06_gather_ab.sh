#!/bin/bash
O=gpurun_out/v29; mkdir -p $O
for rep in 1 2 3; do for v in 1 0; do echo -n "b32 hinted GATHER=$v: " >> $O/ab.txt
  XMEM_GATHER_HIDDEN_INPUT=$v timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/ab.txt; done; done
for rep in 1 2 3; do for v in 0 1; do echo -n "b32 plain GATHER=$v: " >> $O/ab.txt
  XMEM_GATHER_HIDDEN_INPUT=$v timeout 300 python bench.py --scale-only --steps 200 --no-prefetch 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/ab.txt; done; done
cat $O/ab.txt
