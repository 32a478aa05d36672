#!/bin/bash
# A/B of the fp16 filter on the GPU box: isolated per-kernel times of the affinity pipeline (tools/aff_r2_probe.py under
# rocprofv3), optionally after rebuilding affinity_filter.hip with extra -D flags:  filter_ab.sh tag [-DF16_KAPPA=4e-5f ...]
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ $# -gt 0 ]; then
  (cd $R/xmem2_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -x hip "$@" -c affinity_filter.hip -o affinity_filter.o &&
   hipcc --offload-arch=gfx950 -shared -fPIC -o libxmem_hip.so conv_mfma.o elementwise.o affinity.o affinity_filter.o consolidate.o selector.o augment.o)
fi
cd /tmp && export TMPDIR=/tmp
PROBE_FRAMES=${PROBE_FRAMES:-6} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_$tag --output-format csv -- python $R/tools/aff_r2_probe.py > $R/gpurun_out/ab_$tag.log 2>&1 || true
grep -h "hint \|hinted" $R/gpurun_out/ab_$tag.log | tail -4
f=$(find $R/gpurun_out/ab_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag: kernel stats (affinity)"; grep -i "affinity" $f | cut -d, -f1-5 | cut -c1-160
