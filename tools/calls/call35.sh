#!/bin/bash
# comment-only edit of affinity_filter.hip changed the source digest: the PMC passes again (bench.py quotes them by digest), after a
# bit-identity check of the readout
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c35 gpurun_out/prof_r04
export TMPDIR=/tmp
REPO=$PWD
timeout 120 python -m pytest tests/test_gpu_affinity_served_sizes.py -q -x -m gpu -k "B32" 2>&1 | tail -1
( cd /tmp && timeout 200 python $REPO/tools/pmc_bench.py --workload b32 --steps 30 --out $REPO/gpurun_out/prof_r04/r04_bench_b32_pmc_per_frame.json > $REPO/gpurun_out/c35/pmc.log 2>&1 )
tail -2 gpurun_out/c35/pmc.log
python3 -c "
import json; j=json.load(open('gpurun_out/prof_r04/r04_bench_b32_pmc_per_frame.json')); print(j.get('source_digest'), {k:round(v/1e6,1) for k,v in j['families'].items()})"
