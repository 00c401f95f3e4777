#!/bin/bash
# round 3, step j: cin_dw_bf3_k with the two waves of a SIMD half a step out of phase
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x -k "cin or CIN or xdeepfm" > gpurun_out/r03_j_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_j_tests.log
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03j_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03j_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
b cin_backward --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 10 --warmup 2 --no-cpu-baseline
DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh cin_backward -- --workload cin_backward --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_cin_backward.txt 2>&1; echo "== cin_backward"; head -5 gpurun_out/prof_cin_backward.txt | cut -c1-150
