# GPU call 2: streaming GEMM kernel - correctness vs the direct plan and timing of every variant; library-level packed-f32 probe
mkdir -p gpurun_out/c2 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2
S1="1 120 216 256 256"; S2="1 60 108 512 512"; S3="1 60 108 512 256"; S4="1 30 54 576 512"; S5="4 120 216 64 64"; S6="4 60 108 128 128"; S7="4 30 54 256 256"; S8="1 30 54 512 512"; S9="4 30 54 1024 512"; S10="1 60 108 256 256"
F4="19,23,24,25,26,27,28"
for sh in "$S1" "$S2" "$S3" "$S4" "$S5" "$S6" "$S7" "$S8" "$S9" "$S10"; do
  timeout 120 tools/conv_bench -n 30 "$sh" $F4 >> $O/conv_f4.txt 2>&1 || echo "FAILED/timeout on $sh" >> $O/conv_f4.txt
done
# with the fused epilogue flags of the real layers (residual + relu) and relu-on-load
timeout 120 tools/conv_bench -n 20 -r 1,1,1 "$S1" 19,23 "$S7" 19,23 >> $O/conv_f4.txt 2>&1
# F(2x2) family on the 1/16-resolution shapes
for sh in "1 30 54 256 256" "1 30 54 512 256" "1 30 54 512 512" "4 30 54 128 128"; do
  timeout 120 tools/conv_bench -n 30 "$sh" 9,29,30,32,33 >> $O/conv_f2.txt 2>&1 || echo "FAILED/timeout on $sh" >> $O/conv_f2.txt
done
# pointwise layers (k = 1): key-encoder bottlenecks at batch 4 and decoder / value-encoder 1x1s at batch 1
for sh in "4 120 216 64 256 1" "4 120 216 256 64 1" "4 60 108 128 512 1" "4 60 108 512 128 1" "4 30 54 256 1024 1" "4 30 54 1024 256 1" "4 120 216 256 512 1 2" "1 30 54 1024 256 1" "1 30 54 256 1024 1" "1 120 216 256 64 1"; do
  timeout 120 tools/conv_bench -n 30 -r 1,1,1 "$sh" 3,6,2,35,36,38,39 >> $O/conv_1x1.txt 2>&1 || echo "FAILED/timeout on $sh" >> $O/conv_1x1.txt
done
timeout 120 tools/conv_bench -n 30 -r 0,0,1 "4 120 216 64 64 1" 3,35,38 "1 60 108 512 256 1" 3,6,35,36 >> $O/conv_1x1.txt 2>&1
export TMPDIR=/tmp; cd /tmp
for job in "$S1|19" "$S1|23" "$S1|24" "$S5|19" "$S5|23" "$S2|19" "$S2|23" "$S7|19" "$S7|23"; do
  sh="${job%%|*}"; pl="${job##*|}"; tag=$(echo "$sh$pl" | tr ' ' 'x')
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- $GRAFT_REPO_ROOT/tools/conv_bench -n 30 "$sh" $pl > /tmp/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $sh plan $pl" >> $GRAFT_REPO_ROOT/$O/kernel_split.txt
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py $f >> $GRAFT_REPO_ROOT/$O/kernel_split.txt
done
cd $GRAFT_REPO_ROOT
timeout 300 tools/probes/pk_hazard/lib_probe 80 > $O/lib_probe.txt 2>&1
tail -3 $O/lib_probe.txt; tail -4 $O/conv_f4.txt
