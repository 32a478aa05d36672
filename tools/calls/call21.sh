mkdir -p gpurun_out/c21 && cd $GRAFT_REPO_ROOT
bash tools/collect_r06.sh b > gpurun_out/c21/collect.log 2>&1
tail -4 gpurun_out/c21/collect.log
python - <<'P'
import json
for f in ['b32_fp16_loop','b32_split','c3_fp16_loop','c4_fp16_loop']:
    try:
        d=json.loads(open(f'gpurun_out/prof_r06/r06_bench_{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value'],1), d['steps'], d.get('value_long_window'), d.get('value_no_prefetch'), d['schema'])
    except Exception as e: print(f, 'ERR', e)
P
