#!/bin/bash
# round 3, step s: ESMM's towers with their lookups inside the tower kernel
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_tower.py tests/test_gpu_models.py tests/test_gpu_ref_text.py -q -x > gpurun_out/r03_s_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03_s_tests.log
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03s_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03s_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms')" 2>&1)"; }
for r in 1 2; do
b esmm_full --workload esmm_full --steps 50 --warmup 5 --no-cpu-baseline
DIR_TOWER_GATHER=0 b esmm_full_two --workload esmm_full --steps 50 --warmup 5 --no-cpu-baseline
done
b deepfm_full --workload deepfm_full --steps 50 --warmup 10 --no-cpu-baseline
