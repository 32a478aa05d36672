#!/bin/bash
mkdir -p gpurun_out/c25
O=gpurun_out/c25/pointwise_batch.txt
for B in 4 2 1; do
  tools/conv_bench -n 30 -r 0,1,1 "$B 120 216 64 256 1" 3,2 >> $O 2>&1
  tools/conv_bench -n 30 -r 0,0,1 "$B 120 216 256 64 1" 3,2 >> $O 2>&1
  tools/conv_bench -n 30 -r 0,0,1 "$B 120 216 64 64" 23,19 >> $O 2>&1
  tools/conv_bench -n 30 -r 0,1,1 "$B 60 108 128 512 1" 3,2 >> $O 2>&1
  tools/conv_bench -n 30 -r 0,0,1 "$B 60 108 512 128 1" 3,2 >> $O 2>&1
done
cat $O
