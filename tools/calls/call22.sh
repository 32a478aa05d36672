mkdir -p gpurun_out/c22 && cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c22/driver_like.json 2> gpurun_out/c22/driver_like.err ) 2> gpurun_out/c22/time.txt
python - <<'P'
import json
d=json.loads(open('gpurun_out/c22/driver_like.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','steps','warmup','value_long_window','value_no_prefetch','schema')})
print('roofline', {k:d['roofline'].get(k) for k in ('frac','achieved','traffic','call_frac')}, 'alone', d['roofline'].get('alone',{}).get('frac'))
print('conv', d['conv_roofline'].get('frac'), 'cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('threads'), d.get('speedup_vs_cpu'))
print('fp32x', d['value_fp32x'].get('value'), d['value_fp32x'].get('value_long_window'), 'fp16', d['value_fp16_loop'].get('value'), d['value_fp16_loop'].get('value_long_window'))
P
cat gpurun_out/c22/time.txt
