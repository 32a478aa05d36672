# the other workload lines re-collected at schema 6 (self-contained timed region)
mkdir -p gpurun_out/c19 && cd $GRAFT_REPO_ROOT
bash tools/collect_r06.sh c > gpurun_out/c19/collect.log 2>&1
tail -8 gpurun_out/c19/collect.log
python - <<'P'
import json
for f in ['b32dyn','c3','b32motion','c4','c5']:
    try:
        d=json.loads(open(f'gpurun_out/prof_r06/r06_bench_{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value'],1), d['steps'], d.get('value_long_window'), d.get('value_no_prefetch'), d['schema'])
    except Exception as e: print(f, 'ERR', e)
P
