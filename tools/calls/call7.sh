# GPU call 7: the fp16 loop (kernels, stages, end to end), the reworked multi-object gate, fp16 bench line
mkdir -p gpurun_out/c7 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c7
python -m pytest tests/test_gpu_fp16_loop.py -q -s > $O/pytest_fp16.log 2>&1; tail -15 $O/pytest_fp16.log
python -m pytest tests/test_gpu_e2e.py -x -q -s -k "three_objects or noise_floor" > $O/pytest_c3.log 2>&1; tail -4 $O/pytest_c3.log
python -m pytest tests/test_gpu_network.py -x -q > $O/pytest_net.log 2>&1; tail -3 $O/pytest_net.log
python bench.py --precision fp16 --steps 200 --no-cpu-baseline --no-kernel-trace > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -c 400 $O/bench_fp16.err
python - <<'PY'
import json
try:
    j=json.loads(open('gpurun_out/c7/bench_fp16.json').read().strip().splitlines()[-1]); print('fp16 loop fps', j['value'], 'no_prefetch', j.get('value_no_prefetch'))
except Exception as e: print('bench failed', e)
PY
