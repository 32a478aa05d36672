#!/bin/bash
# full GPU suite + smoke at the final state
O=gpurun_out/v32; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests.out 2>&1; echo "tests rc=$?" > $O/summary.txt
tail -3 $O/tests.out >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" >> $O/summary.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_20.json 2> $O/bench20.err; python -c "
import json; d=json.loads(open('$O/bench_default_20.json').read().strip().splitlines()[-1]); print('bench 20 steps', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])" >> $O/summary.txt
cat $O/summary.txt | tail -8
