#!/bin/bash
# round 3, step q: DeepFM inference in one launch (lookups inside the tower kernel)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_tower.py tests/test_gpu_models.py -q -x -k "tower or deepfm or DeepFM" > gpurun_out/r03_q_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03_q_tests.log
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03q_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03q_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
for r in 1 2; do
b deepfm_full --workload deepfm_full --steps 50 --warmup 10 --no-cpu-baseline
DIR_TOWER_GATHER=0 b deepfm_full_two --workload deepfm_full --steps 50 --warmup 10 --no-cpu-baseline
done
DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh deepfm_full -- --workload deepfm_full --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_deepfm_full.txt 2>&1; head -4 gpurun_out/prof_deepfm_full.txt | cut -c1-140
