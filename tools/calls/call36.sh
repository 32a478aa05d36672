#!/bin/bash
mkdir -p gpurun_out/c36
O=gpurun_out/c36/pointwise_stream.txt
P=3,35,38
tools/conv_bench -n 30 -r 0,1,1 "4 120 216 64 256 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,0 "4 120 216 64 256 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 120 216 256 64 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 120 216 64 64 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 60 108 128 512 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 60 108 512 128 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 30 54 256 1024 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 30 54 1024 256 1" $P >> $O 2>&1
grep "^shape" $O | awk '{print $2,$3,$4,$5,$6,$7,$8,"plan",$10,$11,"us",$13,"TF",$NF}'
timeout 200 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "streaming or every_plan or pointwise" 2>&1 | tail -3
