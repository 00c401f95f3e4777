#!/bin/bash
# tools/r02_round.sh (GPU box): the round-2 bench lines and kernel-trace summaries that profiles/ holds -> gpurun_out/
cd "$GRAFT_REPO_ROOT"
b() {  # b NAME args...: one bench.py line -> gpurun_out/r02_bench_NAME.json
    name=$1; shift
    python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r02_bench_$name.json
    echo "$name: $(cut -c1-330 gpurun_out/r02_bench_$name.json)"
}
b default --steps 200 --warmup 20
b din --workload din --steps 50 --warmup 5 --no-cpu-baseline
b din_train --workload din_train --steps 20 --warmup 3 --no-cpu-baseline
b train_sparse --workload train_sparse --steps 50 --warmup 5 --no-cpu-baseline
b train_sparse_zipf --workload train_sparse --id-dist zipf --steps 50 --warmup 5 --no-cpu-baseline
b train_sparse_split --workload train_sparse --train-layout split --steps 50 --warmup 5 --no-cpu-baseline
b deepfm_train --workload deepfm_train --steps 20 --warmup 3 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline
b xdeepfm_full --workload xdeepfm_full --steps 10 --warmup 3 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 5 --warmup 2 --no-cpu-baseline
b cin_backward --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline
b sharded_1gpu --workload sharded_1gpu --steps 50 --warmup 5 --no-cpu-baseline
python3 tools/sweep_shapes.py > gpurun_out/r02_sweep_shapes.md 2> gpurun_out/sweep.err; tail -3 gpurun_out/sweep.err
bash tools/prof.sh default -- --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prof_default.txt 2>&1
bash tools/prof.sh din_train -- --workload din_train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_din_train.txt 2>&1
bash tools/prof.sh train_sparse -- --workload train_sparse --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_train_sparse.txt 2>&1
head -12 gpurun_out/prof_din_train.txt gpurun_out/prof_train_sparse.txt
