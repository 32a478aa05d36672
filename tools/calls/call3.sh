# GPU call 3: what bounds the position GEMMs?  debug knobs (same-operand loads, no stores) + SQ / TCC counters per plan
mkdir -p gpurun_out/c3 && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c3
S1="1 120 216 256 256"; S5="4 120 216 64 64"; S9="4 30 54 1024 512"; S2="1 60 108 512 512"
export TMPDIR=/tmp; cd /tmp
kt() {  # kernel time of the GEMM inside plan $2 on shape $1 (env already set)
  rm -rf /tmp/kt; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $GRAFT_REPO_ROOT/tools/conv_bench -n 30 "$1" $2 > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/kstats.py $f | grep -v "wino4_\|false, false, false"
}
for sh in "$S1" "$S5" "$S9" "$S2"; do
  for pl in 23 26 25; do
    for dbg in 0 1 2 3; do
      echo "== shape $sh plan $pl XMEM_STREAM_DBG=$dbg" >> $O/dbg.txt
      XMEM_STREAM_DBG=$dbg kt "$sh" $pl >> $O/dbg.txt
    done
  done
done
K19='conv_mfma_kernel<64, 64, 1, 1, 32, false, true, false>'
KS='gemm_stream_kernel'
for sh in "$S1" "$S5" "$S9"; do
  for pl in 19 23 26 25; do
    echo "== shape $sh plan $pl" >> $O/pmc.txt
    K="$KS"; [ $pl = 19 ] && K="$K19"
    for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
      rm -rf /tmp/pm
      timeout 120 rocprofv3 --pmc $pass --output-format csv -d /tmp/pm -- $GRAFT_REPO_ROOT/tools/conv_bench -n 10 "$sh" $pl > /tmp/pm.log 2>&1
      f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
      if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py $f "$K" >> $O/pmc.txt; else echo "pass [$pass] failed: $(tail -1 /tmp/pm.log | cut -c1-160)" >> $O/pmc.txt; fi
    done
  done
done
tail -5 $O/pmc.txt
