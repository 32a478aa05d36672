#!/bin/bash
mkdir -p gpurun_out/c28
O=gpurun_out/c28/pointwise_k64.txt
P=3,6,5,2
tools/conv_bench -n 30 -r 0,1,1 "4 120 216 64 256 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,0 "4 120 216 64 256 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "4 120 216 64 64 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,1 "1 120 216 64 64 1" $P >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 60 108 128 512 1" $P >> $O 2>&1
cat $O
