#!/bin/bash
# tools/r02_models_round.sh (GPU box): whole-model bench lines after the bf16x3 dense kernel -> gpurun_out/
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r02_bench_$name.json; echo "$name: $(cut -c1-230 gpurun_out/r02_bench_$name.json)"; }
b deepfm_full --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline
DIR_DENSE_ARITH=f32 b deepfm_full_f32dense --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline
b deepfm_train --workload deepfm_train --steps 20 --warmup 3 --no-cpu-baseline
DIR_DENSE_ARITH=f32 b deepfm_train_f32dense --workload deepfm_train --steps 20 --warmup 3 --no-cpu-baseline
b dcn_full --workload dcn_full --steps 20 --warmup 3 --no-cpu-baseline
DIR_DENSE_ARITH=f32 b dcn_full_f32dense --workload dcn_full --steps 20 --warmup 3 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
b esmm_full --workload esmm_full --steps 50 --warmup 5 --no-cpu-baseline
b esmm_train --workload esmm_train --steps 20 --warmup 3 --no-cpu-baseline
b mlp_dense --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
b xdeepfm_full --workload xdeepfm_full --steps 10 --warmup 3 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 5 --warmup 2 --no-cpu-baseline
bash tools/prof.sh deepfm_full -- --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_deepfm_full.txt 2>&1
head -8 gpurun_out/prof_deepfm_full.txt
