#!/bin/bash
# is it the drains that make the filter contention-sensitive in the three-stream schedule?  tools build, XMEM_F16_DBG=16 (pushes, no drain:
# the lists stay empty, the masks are wrong - timing of the filter kernel only) against the shipped path, same box, traced timed region
O=gpurun_out/v33; mkdir -p $O; export TMPDIR=/tmp
XMEM_HIPCC_FLAGS=-DXMEM_TOOLS timeout 600 python -m xmem2_amd.build --force > $O/build_tools.log 2>&1
for d in 0 16 0 16; do
  XMEM_F16_DBG=$d timeout 600 python bench.py --no-cpu-baseline --no-extra-modes --plain-steps 0 > $O/bench_d$d.json 2> $O/bench_d$d.err
  python - $d <<'P' >> $O/ab.txt
import json,sys
v=sys.argv[1]
d=json.loads(open(f'gpurun_out/v33/bench_d{v}.json').read().strip().splitlines()[-1])
r=d['roofline']; print('XMEM_F16_DBG', v, 'value', round(d['value'],1), {k.split('(')[0][-40:]:(round(x['avg_us'],1), round(x.get('median_us',0),1)) for k,x in (r.get('kernels') or {}).items()})
P
done
cat $O/ab.txt
