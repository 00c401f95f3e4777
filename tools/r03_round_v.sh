#!/bin/bash
# round 3, step v: full GPU suite, smoke, the driver's bench command, and the bench lines the model-level fusions touched
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r03_gpu_suite.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r03_smoke.log
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
b default --steps 200 --warmup 20
b dcn_cross --workload dcn_cross --steps 100 --warmup 10 --no-cpu-baseline
b dcn_cross_backward --workload dcn_cross_backward --steps 100 --warmup 10 --no-cpu-baseline
b mlp_dense --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
DIR_BENCH_DENSE=layers b mlp_dense_layers --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 50 --warmup 10 --no-cpu-baseline
b dcn_full --workload dcn_full --steps 50 --warmup 5 --no-cpu-baseline
b esmm_full --workload esmm_full --steps 50 --warmup 5 --no-cpu-baseline
b xdeepfm_full --workload xdeepfm_full --steps 10 --warmup 2 --no-cpu-baseline
b deepfm_train --workload deepfm_train --steps 30 --warmup 5 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
b esmm_train --workload esmm_train --steps 30 --warmup 5 --no-cpu-baseline
for w in deepfm_full esmm_full dcn_full; do
    DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh $w -- --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1; echo "== $w"; head -4 gpurun_out/prof_$w.txt | cut -c1-150
done
