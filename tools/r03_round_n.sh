#!/bin/bash
# round 3, step n: run-to-run bitwise equality of every bf16x3 kernel at full size (they all hold 16x16x32 MFMAs; see isa_check.py)
cd "$GRAFT_REPO_ROOT"
for t in dense_bf3_stress cin_bf3_stress dense_dw_stress; do
  timeout -k 10 400 python3 tools/$t.py > gpurun_out/r03_stress_$t.txt 2>&1; echo "== $t rc=$?"; grep -v amdgpu.ids gpurun_out/r03_stress_$t.txt | tail -8 | cut -c1-220
done
timeout -k 10 300 python3 tools/tower_stress.py 200 > gpurun_out/r03_stress_tower.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03_stress_tower.txt | tail -3
timeout -k 10 300 python3 tools/din_bf3_stress.py 200 > gpurun_out/r03_stress_din.txt 2>&1; grep "^lib" gpurun_out/r03_stress_din.txt
