#!/bin/bash
# round 3, step o: non-temporal record stores / loads in the DIN training kernels, A/B on one box
cd "$GRAFT_REPO_ROOT"
L=$PWD/details-in-recommendation_amd
for r in 1 2; do
for n in 0 1; do
  DIR_HIP_LIBRARY=$L/libdir_hip_nt$n.so timeout -k 10 300 python3 bench.py --workload din_train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_din_train_nt$n.log 2>&1; echo "din_train nt$n: $(grep '^{' gpurun_out/bench_din_train_nt$n.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
done
done
for n in 0 1; do
  DIR_HIP_LIBRARY=$L/libdir_hip_nt$n.so DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh din_train_nt$n -- --workload din_train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_din_train_nt$n.txt 2>&1; echo "== nt$n"; head -3 gpurun_out/prof_din_train_nt$n.txt | cut -c1-140
done
DIR_HIP_LIBRARY=$L/libdir_hip_nt1.so timeout -k 10 600 python3 -m pytest tests/test_gpu_backward.py -q -x -k "din" > gpurun_out/r03_o_tests.log 2>&1; tail -2 gpurun_out/r03_o_tests.log
