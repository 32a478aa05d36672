#!/bin/bash
# SQ counters of the 64x64 single-accumulator GEMM (conv_mfma_kernel<64,64,1,1,32,false,true,false>) inside F(4x4) layers of the
# batched key encoder, one layer shape per run, separate --pmc passes:  bash tools/probes/conv_counters.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K='conv_mfma_kernel<64, 64, 1, 1, 32, false, true, false>'
for shape in "120 214 64 64" "60 107 128 128" "30 54 256 256"; do
  echo "== batch 4, H W Cin Cout = $shape, plan (19, 1)"
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    rm -rf /tmp/pm
    ONE_CONV_BATCH=4 timeout 120 rocprofv3 --pmc $pass --output-format csv -d /tmp/pm -- python $R/tools/one_conv.py $shape 19 1 20 > /tmp/pm.log 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python $R/tools/pmc_kernel.py $f "$K"; else echo "pass [$pass] failed: $(tail -1 /tmp/pm.log | cut -c1-160)"; fi
  done
done
