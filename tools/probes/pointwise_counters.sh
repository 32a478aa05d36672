#!/bin/bash
# SQ / TCC counters of the classic 64x64 tile on two pointwise layers of the batch-4 key encoder at 1/4 resolution (tools/conv_bench,
# plan 3), separate --pmc passes:  bash tools/probes/pointwise_counters.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K='conv_mfma_kernel<64, 64, 1, 1, 32, false, true'
run() {   # $1 = flags, $2 = shape
  echo "== conv_bench -r $1 \"$2\" plan 3"
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pm
    timeout 60 rocprofv3 --pmc $pass --output-format csv -d /tmp/pm -- $R/tools/conv_bench -n 20 -r $1 "$2" 3 > /tmp/pm.log 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 $R/tools/pmc_kernel.py $f "$K"; else echo "pass [$pass] failed: $(tail -1 /tmp/pm.log | cut -c1-160)"; fi
  done
  grep "^shape" /tmp/pm.log
}
run 0,0,0 "4 120 216 64 256 1"
run 0,0,1 "4 120 216 256 64 1"
