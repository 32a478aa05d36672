cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
python -m pytest tests -q -m gpu -x > gpurun_out/c12/pytest_gpu.log 2>&1; tail -3 gpurun_out/c12/pytest_gpu.log
bash tools/collect_r04.sh a > gpurun_out/collect_a.log 2>&1
bash tools/collect_r04.sh d > gpurun_out/collect_d.log 2>&1
R=r04; OUT=gpurun_out/prof_r04
timeout 900 python bench.py --workload c3 --steps 100 --trace-steps 30 --cpu-frames 6 > $OUT/${R}_bench_c3.json 2> $OUT/bench_c3.err
timeout 900 python bench.py --workload c3 --precision fp16 --steps 100 --no-kernel-trace --cpu-frames 6 > $OUT/${R}_bench_c3_fp16_loop.json 2> $OUT/bench_c3_fp16.err
timeout 600 python bench.py --precision fp16 --keep-trace $OUT/fp16 --cpu-frames 6 > $OUT/${R}_bench_b32_fp16_loop.json 2> $OUT/bench_fp16.err
python tools/trace_table.py $OUT/fp16/b32_kernel_trace.csv > $OUT/${R}_bench_b32_fp16_loop_timed_region_per_frame.csv 2>> $OUT/stats.err
head -12 $OUT/${R}_conv_bench_isolated_layers.txt
python3 - <<PY
import json
for f in ("r04_bench_b32","r04_bench_c3","r04_bench_c3_fp16_loop","r04_bench_b32_fp16_loop"):
    j=json.loads(open("gpurun_out/prof_r04/"+f+".json").read().strip().splitlines()[-1]); p=j.get("parity") or {}
    print(f, round(j["value"],1), {k:p.get(k) for k in ("mask_iou_vs_cpu_min","argmax_mismatch_pixels","argmax_mismatch_pixels_at_clear_cpu_margin","cpu_pixels_near_tie_fraction","max_abs_prob_err_ds8")}, j["roofline"].get("traffic"))
PY
