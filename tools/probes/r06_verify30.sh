#!/bin/bash
O=gpurun_out/v30; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --no-cpu-baseline --no-extra-modes --plain-steps 0 > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/v30/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], r['frac'], r.get('call_frac')); print(json.dumps(r.get('alone'))[:600])
P
tail -3 $O/bench.err
