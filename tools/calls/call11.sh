mkdir -p gpurun_out/c11 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c11
python -m pytest tests/test_gpu_fp16_loop.py -q -k "conv2d_half" > $O/pytest_half.log 2>&1; tail -3 $O/pytest_half.log
cp xmem2_amd/conv_plans_fp16.json $O/conv_plans_fp16_before.json
rm -f xmem2_amd/conv_plans_fp16.json
XMEM_PRECISION=fp16 timeout 900 python tools/tune_convs.py $O/conv_plans_fp16.json > $O/tune_fp16.log 2>&1; tail -1 $O/tune_fp16.log
cp $O/conv_plans_fp16.json xmem2_amd/conv_plans_fp16.json
python bench.py --precision fp16 --steps 200 --no-cpu-baseline --no-kernel-trace --plain-steps 0 > $O/bench_fp16.json 2> $O/bench_fp16.err
python bench.py --precision fp16 --steps 200 --no-cpu-baseline --no-kernel-trace --plain-steps 0 > $O/bench_fp16_b.json 2>> $O/bench_fp16.err
cp $O/conv_plans_fp16_before.json xmem2_amd/conv_plans_fp16.json
python bench.py --precision fp16 --steps 200 --no-cpu-baseline --no-kernel-trace --plain-steps 0 > $O/bench_fp16_old.json 2>> $O/bench_fp16.err
python - <<'PY'
import json
for f in ('bench_fp16','bench_fp16_b','bench_fp16_old'):
    try:
        j=json.loads(open(f'gpurun_out/c11/{f}.json').read().strip().splitlines()[-1]); print(f, round(j['value'],1))
    except Exception as e: print(f,'failed',e)
from collections import Counter
p=json.load(open('gpurun_out/c11/conv_plans_fp16.json')); print(Counter(tuple(v) for v in p.values()).most_common(14))
PY
