#!/bin/bash
# SQ counters of the pass-1 filter kernel at the 720p size (N = 921 600, HW = 3 600; tools/probes/filter_sizes.py c4: 1 un-hinted +
# 12 hinted calls), separate --pmc passes:  bash tools/probes/filter_counters.sh [size]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SZ=${1:-c4}
K='affinity_filter16_kernel<false'
echo "== filter_sizes.py $SZ, kernel $K"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pm
  PROBE_NOCHECK=1 timeout 150 rocprofv3 --pmc $pass --output-format csv -d /tmp/pm -- python $R/tools/probes/filter_sizes.py $SZ > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_kernel.py $f "$K"; else echo "pass [$pass] failed: $(tail -1 /tmp/pm.log | cut -c1-160)"; fi
done
grep "filter pass" /tmp/pm.log
