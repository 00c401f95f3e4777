#!/bin/bash
# tools/r06_tower_lines.sh (GPU box): the tower_bf3_k ("rows") legs of the sweep beside the default kernel's, one box
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r06_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r06_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3))" 2>&1)"; }
b mlp_dense --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_KERNEL=rows b mlp_dense_rows --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_KERNEL=rows DIR_TOWER_RT=2 b mlp_dense_rt2 --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_KERNEL=rows b deepfm_full_rows --workload deepfm_full --steps 200 --warmup 1000 --no-cpu-baseline
export ROUND=r06
bash tools/pmc.sh tower tower_cs_k -- --workload mlp_dense --steps 5 --warmup 1 --no-cpu-baseline
DIR_TOWER_KERNEL=rows bash tools/pmc.sh tower_rows tower_bf3_k -- --workload mlp_dense --steps 5 --warmup 1 --no-cpu-baseline
