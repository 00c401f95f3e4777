#!/bin/bash
set -o pipefail
L=$PWD/details-in-recommendation_amd
for v in ${VARIANTS}; do
  DIR_HIP_LIBRARY=$L/libdir_hip_$v.so timeout -k 10 200 python3 tools/din_c_probe.py > gpurun_out/r03_din_c_probe_$v.txt 2>&1 || echo "$v failed"
  echo "$v: $(grep 'samples with any differing' gpurun_out/r03_din_c_probe_$v.txt)"
done
