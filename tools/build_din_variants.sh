#!/bin/bash
# A/B builds of din_wave.hip (-DDW_EXP=n) linked with the other objects of the in-tree build into libdir_hip_e<n>.so (development only)
set -e
cd "$(dirname "$0")/../details-in-recommendation_amd"
python3 build.py > /dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -Wall -Wno-unused-function \
      -mllvm -amdgpu-atomic-optimizer-strategy=None -DDW_EXP=$n $EXTRA -c csrc/din_wave.hip -o csrc/_build/din_wave_e$n$SUFFIX.o
  objs=$(ls csrc/_build/*.o | grep -v "din_wave")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdir_hip_e$n$SUFFIX.so $objs csrc/_build/din_wave_e$n$SUFFIX.o
  echo "built libdir_hip_e$n$SUFFIX.so"
done
