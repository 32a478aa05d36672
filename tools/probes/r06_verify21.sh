#!/bin/bash
# full GPU suite + headline after the filter emits its lists itself
O=gpurun_out/v21; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests.out 2>&1; echo "tests rc=$?" > $O/summary.txt
tail -4 $O/tests.out >> $O/summary.txt
for i in 1 2; do echo -n "b32 fp32: " >> $O/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt; done
echo -n "b32 no-prefetch: " >> $O/summary.txt
timeout 300 python bench.py --scale-only --steps 200 --no-prefetch 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt
for w in c4 c5; do echo -n "$w: " >> $O/summary.txt
  timeout 600 python bench.py --scale-only --workload $w 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/summary.txt; done
cat $O/summary.txt
