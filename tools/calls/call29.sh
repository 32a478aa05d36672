#!/bin/bash
# full GPU suite + a quick B32 line after the filter-kernel / pointwise changes
mkdir -p gpurun_out/c29
python -m pytest tests -q -m gpu -x > gpurun_out/c29/pytest_gpu.log 2>&1; tail -3 gpurun_out/c29/pytest_gpu.log
timeout 600 python bench.py --no-extra-modes --cpu-frames 4 > gpurun_out/c29/bench_b32.json 2> gpurun_out/c29/bench_b32.err; tail -c 600 gpurun_out/c29/bench_b32.err
python3 - <<PY
import json
j=json.loads(open("gpurun_out/c29/bench_b32.json").read().strip().splitlines()[-1])
print(j["value"], j.get("value_no_prefetch"), j["roofline"], j["conv_roofline"])
PY
