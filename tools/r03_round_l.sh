#!/bin/bash
# round 3, step l: MFMA chains kept dependent so that the SIMD partner's VALU work gets issue slots -- cin_dw_bf3_k (with the stagger) and the DIN forward
cd "$GRAFT_REPO_ROOT"
L=$PWD/details-in-recommendation_amd
DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh cin_backward -- --workload cin_backward --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_cin_backward.txt 2>&1; head -3 gpurun_out/prof_cin_backward.txt | cut -c1-150
for v in chain0 chain1; do
  for r in 1 2; do
  DIR_HIP_LIBRARY=$L/libdir_hip_e0$v.so timeout -k 10 300 python3 bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_din_$v.log 2>&1; echo "din $v: $(grep '^{' gpurun_out/bench_din_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
  done
done
DIR_HIP_LIBRARY=$L/libdir_hip_e0chain1.so timeout -k 10 300 python3 tools/din_bf3_stress.py 50 > gpurun_out/r03_din_stress_chain1.txt 2>&1; grep "^lib" gpurun_out/r03_din_stress_chain1.txt
