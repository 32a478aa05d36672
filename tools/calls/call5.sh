# GPU call 5: B32 before / after the streaming-GEMM plans (same box), plans chosen by the targeted tuner
mkdir -p gpurun_out/c5 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5
python bench.py --steps 200 --no-cpu-baseline --no-kernel-trace --plain-steps 0 > $O/bench_before.json 2> $O/bench_before.err
XMEM_TUNE_STREAM=1 timeout 900 python tools/tune_convs.py $O/conv_plans.json > $O/tune.log 2>&1
tail -3 $O/tune.log
cp xmem2_amd/conv_plans.json $O/conv_plans_before.json
cp $O/conv_plans.json xmem2_amd/conv_plans.json
python bench.py --steps 200 --no-cpu-baseline --no-kernel-trace --plain-steps 0 > $O/bench_after.json 2> $O/bench_after.err
python bench.py --steps 200 --no-cpu-baseline --no-kernel-trace --plain-steps 0 > $O/bench_after2.json 2> $O/bench_after2.err
python - <<'PY'
import json
for f in ('before','after','after2'):
    try:
        j=json.loads(open(f'gpurun_out/c5/bench_{f}.json').read().strip().splitlines()[-1]); print(f, round(j['value'],1), 'fps', round(j['ms_per_step'],4),'ms')
    except Exception as e: print(f, 'failed', e)
PY
