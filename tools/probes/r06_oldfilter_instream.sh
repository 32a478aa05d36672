#!/bin/bash
O=gpurun_out/v34; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-extra-modes --plain-steps 0 > $O/bench_$i.json 2> $O/bench_$i.err
  python - $i <<'P' >> $O/ab.txt
import json,sys
v=sys.argv[1]
d=json.loads(open(f'gpurun_out/v34/bench_{v}.json').read().strip().splitlines()[-1])
r=d['roofline']; print('run', v, 'value', round(d['value'],1), 'frac', round(r['frac'],3), {k.split('(')[0][-36:]:(round(x['avg_us'],1), round(x.get('median_us',0),1)) for k,x in (r.get('kernels') or {}).items()})
P
done
cat $O/ab.txt
