#!/bin/bash
# the streaming-GEMM variants of the new shapes' Winograd plans (XMEM_TUNE_STREAM=1: bit-identical results either way)
O=gpurun_out/v38; mkdir -p $O
XMEM_TUNE_STREAM=1 XMEM_TUNE_GEOMS="1080x1920x5,1080x1920x2,1080x1920x3,720x1280x2,720x1280x3,480x854x4,480x854x5" timeout 1500 python tools/tune_convs.py $O/conv_plans.json > $O/tune.log 2>&1
tail -3 $O/tune.log | cut -c1-200
cp $O/conv_plans.json xmem2_amd/conv_plans.json
for i in 1 2; do echo -n "c5 with streaming variants: " ; timeout 600 python bench.py --scale-only --workload c5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; done
echo -n "c4: "; timeout 600 python bench.py --scale-only --workload c4 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
echo -n "c3: "; timeout 600 python bench.py --scale-only --workload c3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
echo -n "b32: "; timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
