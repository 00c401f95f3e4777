#!/bin/bash
set -o pipefail
L=$PWD/details-in-recommendation_amd
for m in base zero_hist_rows small_rows; do
  PROBE_MODE=$m DIR_HIP_LIBRARY=$L/libdir_hip_a161.so timeout -k 10 200 python3 tools/din_c_probe.py > gpurun_out/r03_din_c_probe_mode_$m.txt 2>&1 || echo "$m failed"
  echo "$m: $(grep 'samples with any differing' gpurun_out/r03_din_c_probe_mode_$m.txt)"
done
