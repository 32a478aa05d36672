#!/bin/bash
# after the key projection went to the Winograd plans (host-side change: zero filters up to 132 channels): full GPU suite, part a
# (B32 line + trace + stats + PMC) and the config-3 line again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c33
python -m pytest tests -q -m gpu -x > gpurun_out/c33/pytest_gpu.log 2>&1; tail -3 gpurun_out/c33/pytest_gpu.log
bash tools/collect_r04.sh a > gpurun_out/collect_a.log 2>&1
R=r04; OUT=gpurun_out/prof_r04
timeout 900 python bench.py --workload c3 --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_c3.json 2> $OUT/bench_c3.err
python3 - <<PY
import json
for f in ("r04_bench_b32","r04_bench_c3"):
    j=json.loads(open("gpurun_out/prof_r04/"+f+".json").read().strip().splitlines()[-1]); p=j.get("parity") or {}
    print(f, round(j["value"],1), j.get("value_no_prefetch"), (j.get("value_fp32x") or {}).get("value"), (j.get("value_fp16_loop") or {}).get("value"), {k:p.get(k) for k in ("mask_iou_vs_cpu_min","argmax_mismatch_pixels","argmax_mismatch_pixels_at_clear_cpu_margin")}, j["roofline"].get("frac"), j["roofline"].get("frac_median"), j["roofline"].get("traffic"), (j.get("conv_roofline") or {}).get("us_per_frame"), (j.get("conv_roofline") or {}).get("frac"))
PY
