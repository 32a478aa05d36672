#!/bin/bash
# the readout stream at high priority: frames/s and the in-stream durations of the select's kernels
O=gpurun_out/v31; mkdir -p $O; export TMPDIR=/tmp
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > $O/ab.txt 2>&1
for rep in 1 2 3; do for v in 0 -1; do echo -n "b32 hinted XMEM_READOUT_PRIORITY=$v: " >> $O/ab.txt
  XMEM_READOUT_PRIORITY=$v timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> $O/ab.txt; done; done
for v in 0 -1; do
  XMEM_READOUT_PRIORITY=$v timeout 600 python bench.py --no-cpu-baseline --no-extra-modes --plain-steps 0 > $O/bench_p$v.json 2> $O/bench_p$v.err
  python - $v <<'P' >> $O/ab.txt
import json,sys
v=sys.argv[1]
d=json.loads(open(f'gpurun_out/v31/bench_p{v}.json').read().strip().splitlines()[-1])
r=d['roofline']; print('priority', v, 'value', round(d['value'],1), 'frac', round(r['frac'],3), 'call_frac', round(r.get('call_frac') or 0,3), {k.split('(')[0][-40:]:(round(x['avg_us'],1), round(x.get('median_us',0),1)) for k,x in (r.get('kernels') or {}).items()})
P
done
cat $O/ab.txt
