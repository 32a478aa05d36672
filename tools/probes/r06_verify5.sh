#!/bin/bash
mkdir -p gpurun_out/v5
bash tools/probes/r06_decoder_layers.sh > gpurun_out/v5/decoder_layers.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py -q -x -k "conv or wino or network or encode or segment or golden" > gpurun_out/v5/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v5/summary.txt
tail -5 gpurun_out/v5/tests.out >> gpurun_out/v5/summary.txt
for rep in 1 2; do echo -n "b32: " >> gpurun_out/v5/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v5/summary.txt; done
cat gpurun_out/v5/decoder_layers.txt gpurun_out/v5/summary.txt
