#!/bin/bash
mkdir -p gpurun_out/c27
O=gpurun_out/c27/pointwise_plans.txt
P=${PLANS:-3,2,6,35,36,38,39}
tools/conv_bench -n 30 -r 0,1,1 "4 120 216 64 256 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,0 "4 120 216 64 256 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 120 216 256 64 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 120 216 64 64 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 60 108 128 512 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 60 108 512 128 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 30 54 256 1024 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 30 54 1024 256 1" $P >> $O 2>&1
cat $O
