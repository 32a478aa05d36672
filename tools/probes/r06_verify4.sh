#!/bin/bash
mkdir -p gpurun_out/v4
bash tools/probes/r06_decoder_layers.sh > gpurun_out/v4/decoder_layers.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_c5_stream.py "tests/test_gpu_e2e.py::test_e2e_480p_three_objects_bench_c3_stream_vs_float64_reference" "tests/test_gpu_e2e.py::test_e2e_480p_three_objects_plain_checkpoint_noise_floor" -q -s > gpurun_out/v4/tests.out 2>&1; echo "tests rc=$?" > gpurun_out/v4/summary.txt
grep -v "^$" gpurun_out/v4/tests.out | grep -v "^E   " | tail -30 | cut -c1-700 >> gpurun_out/v4/summary.txt
cat gpurun_out/v4/decoder_layers.txt gpurun_out/v4/summary.txt
