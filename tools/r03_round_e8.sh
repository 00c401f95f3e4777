#!/bin/bash
set -o pipefail
L=$PWD/details-in-recommendation_amd
DIR_HIP_LIBRARY=$L/libdir_hip_e161ns.so timeout -k 10 200 python3 tools/din_c_probe.py > gpurun_out/r03_din_c_probe_e161ns.txt 2>&1 || echo "failed"
echo "e161ns: $(grep 'samples with any differing' gpurun_out/r03_din_c_probe_e161ns.txt)"
for n in 1ns 0ns; do
  DIR_HIP_LIBRARY=$L/libdir_hip_e$n.so timeout -k 10 300 python3 tools/din_bf3_stress.py 100 > gpurun_out/r03_din_stress_e$n.txt 2>&1 || echo "e$n failed"
  grep "^lib" gpurun_out/r03_din_stress_e$n.txt
done
