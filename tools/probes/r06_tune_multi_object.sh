#!/bin/bash
# plans for the (resolution, objects) pairs the shipped table does not list: config 5's decoder ran on the built-in heuristic
O=gpurun_out/v37; mkdir -p $O
XMEM_TUNE_GEOMS="1080x1920x5,1080x1920x2,1080x1920x3,720x1280x2,720x1280x3,480x854x4,480x854x5" timeout 1500 python tools/tune_convs.py $O/conv_plans.json > $O/tune.log 2>&1
tail -5 $O/tune.log | cut -c1-200; grep -c "^[0-9]" $O/tune.log
for i in 1; do echo -n "c5 with the shipped table: " ; timeout 600 python bench.py --scale-only --workload c5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; done
cp xmem2_amd/conv_plans.json /tmp/conv_plans.keep; cp $O/conv_plans.json xmem2_amd/conv_plans.json
for i in 1 2; do echo -n "c5 with the extended table: " ; timeout 600 python bench.py --scale-only --workload c5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; done
echo -n "b32 with the extended table: "; timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
