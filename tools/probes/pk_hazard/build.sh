#!/bin/bash
# builds tools/probes/pk_hazard/pk_hazard: the victims twice (packed / unpacked f32 VALU), the driver once
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -c victim_pk.hip -o victim_pk.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops -c victim_nopk.hip -o victim_nopk.o 2>&1 | grep -v "not a recognized feature" || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -c pk_hazard.hip -o pk_hazard.o
hipcc --offload-arch=gfx950 -fgpu-rdc victim_pk.o victim_nopk.o pk_hazard.o -o pk_hazard
echo built $(pwd)/pk_hazard
