#!/bin/bash
set -o pipefail
L=$PWD/details-in-recommendation_amd
for n in 0nopk 1nopk; do
  DIR_HIP_LIBRARY=$L/libdir_hip_e$n.so timeout -k 10 300 python3 tools/din_bf3_stress.py 100 > gpurun_out/r03_din_stress_e$n.txt 2>&1 || echo "e$n failed"
  grep "^lib" gpurun_out/r03_din_stress_e$n.txt
  DIR_DIN_STATIC=1 DIR_HIP_LIBRARY=$L/libdir_hip_e$n.so timeout -k 10 300 python3 tools/din_bf3_stress.py 10 > gpurun_out/r03_din_stress_e${n}_static.txt 2>&1 || echo "e$n static failed"
  echo "static: $(grep '^lib' gpurun_out/r03_din_stress_e${n}_static.txt)"
done
DIR_DIN_ARITH=f32 DIR_HIP_LIBRARY=$L/libdir_hip_e0nopk.so timeout -k 10 300 python3 tools/din_bf3_stress.py 20 > gpurun_out/r03_din_stress_f32nopk.txt 2>&1; grep "^lib" gpurun_out/r03_din_stress_f32nopk.txt
