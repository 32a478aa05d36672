# GPU call 6: new tests + the default bench line with the new roofline fields
mkdir -p gpurun_out/c6 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c6
python -m pytest tests/test_gpu_ops.py -x -q -k "conv or stream" > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_augment.py -x -q -s > $O/pytest_e2e.log 2>&1; tail -3 $O/pytest_e2e.log
export TMPDIR=/tmp
python bench.py --keep-trace $O/trace > $O/bench_b32.json 2> $O/bench_b32.err; tail -c 600 $O/bench_b32.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/c6/bench_b32.json').read().strip().splitlines()[-1])
print('fps', j['value'], 'no_prefetch', j.get('value_no_prefetch'))
r=j['roofline']; print({k:r.get(k) for k in ('source','achieved','frac','frac_median','kernel_avg_us','kernel_median_us','kernel_launches_timed','from_hip_events')})
c=j.get('conv_roofline'); print({k:c.get(k) for k in ('achieved','frac','executed_mfma_gflop_per_frame','algorithmic_tflops','algorithmic_gflop_per_frame','algorithmic_gflop_per_frame_surveyed','us_per_frame')} if c else None)
print(j.get('parity')); print(j.get('cpu_baseline',{}).get('value'))
PY
