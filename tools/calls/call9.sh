# GPU call 9: plans of the fp16 loop's half kernels; its bench line before / after with the kernel trace of the timed region
mkdir -p gpurun_out/c9 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c9
export TMPDIR=/tmp
XMEM_PRECISION=fp16 timeout 900 python tools/tune_convs.py $O/conv_plans_fp16.json > $O/tune_fp16.log 2>&1; tail -2 $O/tune_fp16.log
cp $O/conv_plans_fp16.json xmem2_amd/conv_plans_fp16.json
python bench.py --precision fp16 --steps 200 --no-cpu-baseline --keep-trace $O/trace > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -c 300 $O/bench_fp16.err
python tools/trace_table.py $O/trace/b32_kernel_trace.csv > $O/fp16_timed_region_per_frame.csv 2>> $O/bench_fp16.err
head -40 $O/fp16_timed_region_per_frame.csv
python - <<'PY'
import json
j=json.loads(open('gpurun_out/c9/bench_fp16.json').read().strip().splitlines()[-1]); print('fp16 loop fps', j['value'], 'no_prefetch', j.get('value_no_prefetch'), 'parity', j.get('parity'))
PY
