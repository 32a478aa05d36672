# B32 line + traces re-collected (schema 6; conv survey fixed), PMC passes kept (same kernel digest)
mkdir -p gpurun_out/c17 && cd $GRAFT_REPO_ROOT
SKIP_PMC=1 bash tools/collect_r06.sh a > gpurun_out/c17/collect.log 2>&1
tail -3 gpurun_out/c17/collect.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/prof_r06/r06_bench_b32.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','steps','warmup','value_long_window','value_no_prefetch','schema')})
print('roofline frac',d['roofline'].get('frac'),'alone',d['roofline'].get('alone',{}).get('frac'), 'traffic', d['roofline'].get('traffic'))
print('conv', d['conv_roofline'].get('frac'), d['conv_roofline'].get('us_per_frame'), d['conv_roofline'].get('algorithmic_gflop_per_frame_surveyed'))
P
