#!/bin/bash
# reproducibility of the shipped MFMA kernels that run two waves per SIMD: the fused tower and the fp32 DIN forward
set -o pipefail
timeout -k 10 300 python3 tools/tower_stress.py 200 > gpurun_out/r03_tower_stress.txt 2>&1 || echo "tower stress failed"
tail -3 gpurun_out/r03_tower_stress.txt
DIR_DIN_ARITH=f32 timeout -k 10 300 python3 tools/din_bf3_stress.py 200 > gpurun_out/r03_din_stress_f32.txt 2>&1 || echo "din f32 stress failed"
tail -4 gpurun_out/r03_din_stress_f32.txt
