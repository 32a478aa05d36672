#!/bin/bash
# A/B of the 1x1 loader's operand prefetch depth (XMEM_CONV_PF, csrc/conv_mfma.hip) on the pointwise layers of the batch-4 key encoder,
# against a library built from the previous commit (tools/probes/ab_old/libxmem_hip.so, LD_LIBRARY_PATH wins over the rpath).
# usage: bash tools/probes/ab_pointwise.sh > gpurun_out/.../ab_pointwise.txt
B=tools/conv_bench
run() {  # $1 = label, rest = env
  echo "== $1"
  shift
  env "$@" $B -n 40 -r 0,1,1 "4 120 216 64 256 1" 3,6 "4 60 108 128 512 1" 3,6 "4 30 54 256 1024 1" 3,6
  env "$@" $B -n 40 -r 0,0,1 "4 120 216 256 64 1" 3,6 "4 120 216 64 64 1" 3 "4 120 216 256 128 1" 3,6,2 "4 60 108 512 128 1" 3,6,2 "4 60 108 512 256 1" 3,6,2 "4 30 54 1024 256 1" 3,6,2
  env "$@" $B -n 40 -r 0,0,0 "4 120 216 64 256 1" 3,6 "4 120 216 256 512 1 2" 3,6 "4 60 108 512 1024 1 2" 3,6
  env "$@" $B -n 40 -r 0,0,0 "1 30 54 512 256 1" 3,6 "1 30 54 1024 512 1" 3,6
}
run "previous commit (PF 1, scale/shift fetched in the epilogue)" LD_LIBRARY_PATH=tools/probes/ab_old
run "this build, PF 1 (scale/shift requested at kernel start)" XMEM_CONV_PF=1
run "this build, PF 2" XMEM_CONV_PF=2
run "this build, PF 3" XMEM_CONV_PF=3
run "this build, PF 4" XMEM_CONV_PF=4
