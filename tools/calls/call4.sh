# GPU call 4: strided unit order vs contiguous runs; component knock-outs (barrier / LDS-DMA issue / fragment reads); stage-3 hazard probe
mkdir -p gpurun_out/c4 && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c4
S1="1 120 216 256 256"; S5="4 120 216 64 64"; S9="4 30 54 1024 512"; S2="1 60 108 512 512"; S7="4 30 54 256 256"; S8="1 30 54 512 512"
timeout 300 tools/probes/pk_hazard/pk_hazard2 40 > $O/pk_hazard2.txt 2>&1
for sh in "$S1" "$S2" "$S5" "$S7" "$S8" "$S9"; do
  timeout 120 tools/conv_bench -n 30 "$sh" 19,23,26,24 >> $O/conv_f4.txt 2>&1
  XMEM_STREAM_DBG=32 timeout 120 tools/conv_bench -n 30 "$sh" 23,26 2>&1 | sed 's/$/   [contiguous runs]/' >> $O/conv_f4.txt
done
export TMPDIR=/tmp; cd /tmp
kt() {
  rm -rf /tmp/kt; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $GRAFT_REPO_ROOT/tools/conv_bench -n 30 "$1" $2 > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py $f | grep -v "wino4_\|false, false, false"
}
for sh in "$S1" "$S9" "$S5"; do
  for pl in 23 26; do
    for dbg in 0 32 1 3 4 8 16 20 28 31; do
      echo "== shape $sh plan $pl XMEM_STREAM_DBG=$dbg" >> $O/dbg.txt
      XMEM_STREAM_DBG=$dbg kt "$sh" $pl >> $O/dbg.txt
    done
  done
done
K='gemm_stream_kernel'
for sh in "$S1" "$S9"; do
  for pl in 23 26; do
    echo "== shape $sh plan $pl" >> $O/pmc.txt
    for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
      rm -rf /tmp/pm
      timeout 120 rocprofv3 --pmc $pass --output-format csv -d /tmp/pm -- $GRAFT_REPO_ROOT/tools/conv_bench -n 10 "$sh" $pl > /tmp/pm.log 2>&1
      f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
      if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py $f "$K" >> $O/pmc.txt; else echo "pass [$pass] failed" >> $O/pmc.txt; fi
    done
  done
done
tail -3 $O/pk_hazard2.txt
