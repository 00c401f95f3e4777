#!/bin/bash
set -o pipefail
L=$PWD/details-in-recommendation_amd
DIR_DIN_STATIC=1 DIR_HIP_LIBRARY=$L/libdir_hip_e161.so timeout -k 10 300 python3 tools/din_bf3_stress.py 20 > gpurun_out/r03_din_stress_e161_static.txt 2>&1 || echo failed
grep -v "^  run [1-9]" gpurun_out/r03_din_stress_e161_static.txt | tail -8
DIR_DIN_STATIC=1 DIR_HIP_LIBRARY=$L/libdir_hip_e1.so timeout -k 10 300 python3 tools/din_bf3_stress.py 40 > gpurun_out/r03_din_stress_e1_static.txt 2>&1 || echo failed
grep -v "^  run [1-9]" gpurun_out/r03_din_stress_e1_static.txt | tail -8
