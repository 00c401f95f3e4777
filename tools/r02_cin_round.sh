#!/bin/bash
# tools/r02_cin_round.sh (GPU box): bench lines, kernel-trace summary and PMC pass for the CIN workloads of round 2 -> gpurun_out/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
python3 bench.py --workload cin --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cin_bf3.log 2>&1 && grep '^{' gpurun_out/bench_cin_bf3.log | tail -1 > gpurun_out/r02_bench_cin.json
python3 bench.py --workload cin --cin-arith f32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cin_f32.log 2>&1 && grep '^{' gpurun_out/bench_cin_f32.log | tail -1 > gpurun_out/r02_bench_cin_f32.json
python3 bench.py --workload xdeepfm_full --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_xdfm.log 2>&1 && grep '^{' gpurun_out/bench_xdfm.log | tail -1 > gpurun_out/r02_bench_xdeepfm_full.json
bash tools/prof.sh cin -- --workload cin --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_cin.txt 2>&1
bash tools/pmc.sh cin cin_bf3_k -- --workload cin --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_cin.txt 2>&1
cat gpurun_out/r02_bench_cin.json gpurun_out/r02_bench_cin_f32.json gpurun_out/r02_bench_xdeepfm_full.json | cut -c1-900
cat gpurun_out/prof_cin.txt gpurun_out/pmc_cin.txt
