#!/bin/bash
# DIN bf16x3 reproducibility A/B: e0 = accumulators start at the LDS-loaded per-sample term, e1 = start at zero, term added on the VALU
set -o pipefail
for n in ${VARIANTS:-0 1}; do
  DIR_HIP_LIBRARY=$PWD/details-in-recommendation_amd/libdir_hip_e$n.so timeout -k 10 300 python3 tools/din_bf3_stress.py 40 > gpurun_out/r03_din_stress_e$n.txt 2>&1 || echo "e$n failed"
  tail -12 gpurun_out/r03_din_stress_e$n.txt
done
