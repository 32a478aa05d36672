mkdir -p gpurun_out/c1 && cd $GRAFT_REPO_ROOT
( tools/probes/pk_hazard/pk_hazard 60 > gpurun_out/c1/pk_hazard.txt 2>&1 ) 
S1="1 120 216 256 256"; S2="1 60 108 512 512"; S3="1 60 108 512 256"; S4="1 30 54 576 512"; S5="4 120 216 64 64"; S6="4 60 108 128 128"; S7="4 30 54 256 256"; S8="1 30 54 512 512"; S9="4 30 54 1024 512"
tools/conv_bench -n 30 "$S1" 3,9,19,17,18,14 "$S2" 3,9,19,17,18 "$S3" 3,9,19,18 "$S4" 3,9,19 "$S5" 3,9,19,13,14 "$S6" 3,9,19,14 "$S7" 3,9,19,14 "$S8" 3,9,19 "$S9" 3,9,19 > gpurun_out/c1/conv_bench.txt 2>&1
export TMPDIR=/tmp; cd /tmp
for sh in "$S1" "$S2" "$S5" "$S7"; do
  tag=$(echo $sh | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- $GRAFT_REPO_ROOT/tools/conv_bench -n 30 "$sh" 19 > /tmp/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $sh plan 19" >> $GRAFT_REPO_ROOT/gpurun_out/c1/kernel_split.txt
  [ -n "$f" ] && cut -d, -f1-8 $f | head -8 >> $GRAFT_REPO_ROOT/gpurun_out/c1/kernel_split.txt
done
cd $GRAFT_REPO_ROOT
timeout 900 python tests/parity_by_plan.py > gpurun_out/c1/parity_by_plan.txt 2>&1
tail -5 gpurun_out/c1/parity_by_plan.txt
