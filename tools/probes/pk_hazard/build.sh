#!/bin/bash
# builds tools/probes/pk_hazard/pk_hazard: the victims twice (packed / unpacked f32 VALU), the driver once
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -c victim_pk.hip -o victim_pk.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops -c victim_nopk.hip -o victim_nopk.o 2>&1 | grep -v "not a recognized feature" || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -c pk_hazard.hip -o pk_hazard.o
hipcc --offload-arch=gfx950 -fgpu-rdc victim_pk.o victim_nopk.o pk_hazard.o -o pk_hazard
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops -c pk_hazard2.hip -o pk_hazard2.o 2>&1 | grep -v "not a recognized feature" || true
hipcc --offload-arch=gfx950 -fgpu-rdc victim_pk.o victim_nopk.o pk_hazard2.o -o pk_hazard2
echo built $(pwd)/pk_hazard
if [ "$1" = "lib" ]; then
  # the library twice: the shipped build is xmem2_amd/csrc/libxmem_hip.so; here the same sources WITHOUT -packed-fp32-ops
  R=$(cd ../../.. && pwd)
  T=$(mktemp -d)
  for f in conv_mfma gemm_stream elementwise affinity affinity_filter consolidate selector augment; do
    x=""; [ $f = augment ] && x="-ffp-contract=off"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $x -x hip -c $R/xmem2_amd/csrc/$f.hip -o $T/$f.o &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o libxmem_hip_pk.so $T/*.o
  rm -rf $T libxmem_hip_pk.so.*
  hipcc -O2 -std=c++17 lib_probe.cpp -I $R/include -ldl -o lib_probe
  echo built $(pwd)/lib_probe and libxmem_hip_pk.so
fi
