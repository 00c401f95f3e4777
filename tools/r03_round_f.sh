#!/bin/bash
# round 3, step f: library without packed fp32 VALU instructions in embedding_bag / backward / din_wave, DIN forward on bf16x3:
# DIN tests, then the bench lines that could move
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x -k "din or DIN or bag or gather or cross or fm" > gpurun_out/r03_f_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_f_tests.log
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03f_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03f_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
b default --steps 200 --warmup 20 --no-cpu-baseline
b gather_only --workload gather_only --steps 200 --warmup 20 --no-cpu-baseline
b deepfm_sparse_packed --workload deepfm_sparse_packed --steps 200 --warmup 20 --no-cpu-baseline
b multihot_bag --workload multihot_bag --steps 50 --warmup 5 --no-cpu-baseline
b dcn_cross_backward --workload dcn_cross_backward --steps 100 --warmup 10 --no-cpu-baseline
b din --workload din --steps 50 --warmup 5 --no-cpu-baseline
DIR_DIN_STATIC=0 b din_queue --workload din --steps 50 --warmup 5 --no-cpu-baseline
DIR_DIN_ARITH=f32 b din_f32 --workload din --steps 50 --warmup 5 --no-cpu-baseline
DIR_DIN_ARITH=f32 DIR_DIN_STATIC=1 b din_f32_static --workload din --steps 50 --warmup 5 --no-cpu-baseline
b din_train --workload din_train --steps 30 --warmup 5 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 50 --warmup 10 --no-cpu-baseline
b dcn_full --workload dcn_full --steps 50 --warmup 5 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
b train_sparse --workload train_sparse --steps 100 --warmup 10 --no-cpu-baseline
b small_batch --workload small_batch --steps 200 --warmup 20 --no-cpu-baseline
