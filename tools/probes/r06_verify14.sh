#!/bin/bash
mkdir -p gpurun_out/v14
bash tools/probes/r06_decoder_layers.sh > gpurun_out/v14/decoder_layers.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py tests/test_gpu_fp16_loop.py -q -x > gpurun_out/v14/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v14/summary.txt
tail -3 gpurun_out/v14/tests.out >> gpurun_out/v14/summary.txt
for p in fp32 fp32 fp32 fp16; do echo -n "b32 $p: " >> gpurun_out/v14/summary.txt
  timeout 300 python bench.py --scale-only --steps 200 --precision $p 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])" >> gpurun_out/v14/summary.txt; done
grep "wino\|^shape" gpurun_out/v14/decoder_layers.txt | cut -c1-150; cat gpurun_out/v14/summary.txt
