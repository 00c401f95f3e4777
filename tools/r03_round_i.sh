#!/bin/bash
# round 3, step i: the saved-activation row pass with its two GEMMs on bf16x3 -- tests, then A/B timing
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x -k "din or DIN" > gpurun_out/r03_i_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03_i_tests.log
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03i_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03i_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
b din_train --workload din_train --steps 30 --warmup 5 --no-cpu-baseline
DIR_DIN_BWD_ARITH=f32 b din_train_bwd_f32 --workload din_train --steps 30 --warmup 5 --no-cpu-baseline
DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh din_train -- --workload din_train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_din_train.txt 2>&1; echo "== din_train"; head -4 gpurun_out/prof_din_train.txt | cut -c1-150
