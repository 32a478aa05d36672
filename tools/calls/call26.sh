#!/bin/bash
mkdir -p gpurun_out/c26
O=gpurun_out/c26/pointwise_res.txt
tools/conv_bench -n 30 -r 0,1,1 "4 120 216 64 256 1" 3,2 >> $O 2>&1
tools/conv_bench -n 30 -r 0,0,0 "4 120 216 64 256 1" 3,2 >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 60 108 128 512 1" 3,2 >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "4 30 54 256 1024 1" 3,2 >> $O 2>&1
tools/conv_bench -n 30 -r 0,1,1 "1 120 216 256 256" 3,23 >> $O 2>&1
cat $O
