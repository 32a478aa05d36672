# the driver's own command on the schema-6 bench (self-contained timed region + value_long_window), then the B32 line + traces re-collected
mkdir -p gpurun_out/c16 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c16
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like.json 2> $O/driver_like.err ) 2> $O/driver_like.time
python - <<'P'
import json
d=json.loads(open('gpurun_out/c16/driver_like.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','steps','warmup','value_long_window','value_no_prefetch','schema')})
print('fp32x',d.get('value_fp32x')); print('fp16',d.get('value_fp16_loop'))
print('roofline frac',d['roofline'].get('frac'),'alone',d['roofline'].get('alone',{}).get('frac'))
print('parity', {k:d['parity'].get(k) for k in ('mask_iou_vs_cpu_min','argmax_mismatch_pixels','argmax_mismatch_pixels_at_clear_cpu_margin','argmax_mismatch_pixels_at_survey_margin')})
P
cat $O/driver_like.time
SKIP_PMC=1 bash tools/collect_r06.sh a > $O/collect.log 2>&1
tail -3 $O/collect.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/prof_r06/r06_bench_b32.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','steps','warmup','value_long_window','value_no_prefetch','schema')})
print('fp32x',d.get('value_fp32x')); print('fp16',d.get('value_fp16_loop'))
print('roofline frac',d['roofline'].get('frac'),'alone',d['roofline'].get('alone',{}).get('frac'), 'traffic', d['roofline'].get('traffic'))
print('conv', d['conv_roofline'].get('frac'), d['conv_roofline'].get('us_per_frame'))
P
