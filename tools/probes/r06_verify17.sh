#!/bin/bash
# round 6, step 20b: candidate buffer in the pad bytes of the staged rows (LDS back to 76 KB per workgroup)
O=gpurun_out/v17; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_affinity_served_sizes.py tests/test_gpu_ops.py tests/test_gpu_memory.py -q -x > $O/tests_a.out 2>&1; echo "tests_a rc=$?" > $O/summary.txt
tail -3 $O/tests_a.out >> $O/summary.txt
export TMPDIR=/tmp
timeout 900 python bench.py --keep-trace $PWD/$O > $O/bench_b32.json 2> $O/bench_b32.err
python tools/trace_table.py $O/b32_kernel_trace.csv > $O/bench_b32_timed_region_per_frame.csv 2>> $O/stats.err
rm -f $O/b32_kernel_trace.csv
python - <<'P' >> $O/summary.txt
import json
d=json.loads(open('gpurun_out/v17/bench_b32.json').read().strip().splitlines()[-1])
print('full bench value', d['value'], 'no_prefetch', d.get('value_no_prefetch'))
r=d['roofline']; print('frac', r['frac'], 'call_frac', r.get('call_frac'), 'traffic', r.get('traffic'))
for k,v in (r.get('kernels') or {}).items(): print('  ', k, v['launches_per_frame'], round(v['avg_us'],1), round(v.get('median_us',0),1))
print('families', {k: round(v['us_per_frame'],1) for k,v in d['kernel_trace']['families'].items()})
print('parity', d['parity']['argmax_mismatch_pixels'], d['parity']['mask_iou_vs_cpu_min'])
P
cat $O/summary.txt
