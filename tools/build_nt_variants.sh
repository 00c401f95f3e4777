#!/bin/bash
# A/B builds of the DIN training kernels with / without non-temporal record traffic (-DDIN_NT=0|1) -> libdir_hip_nt<0|1>.so (development only)
set -e
cd "$(dirname "$0")/../details-in-recommendation_amd"
python3 build.py > /dev/null 2>&1
for n in 0 1; do
  for f in din_wave din_bwd_rows; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -Wall -Wno-unused-function \
      -mllvm -amdgpu-atomic-optimizer-strategy=None -Xclang -target-feature -Xclang -packed-fp32-ops -DDIN_NT=$n -c csrc/$f.hip -o csrc/_build/${f}_nt$n.o 2>&1 | grep -v "not a recognized" || true
  done
  objs=$(ls csrc/_build/*.o | grep -v "din_wave\|din_bwd_rows\|_nt[01]\|_e[0-9]")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdir_hip_nt$n.so $objs csrc/_build/din_wave_nt$n.o csrc/_build/din_bwd_rows_nt$n.o
  echo "built libdir_hip_nt$n.so"
done
