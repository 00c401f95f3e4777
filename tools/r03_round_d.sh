#!/bin/bash
# tools/r03_round_d.sh (GPU box): HIP Adam tests, then the DCN training step's bench line and kernel trace
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "adam" > gpurun_out/r03_tests_adam.log 2>&1; echo "adam tests rc=$?"; tail -12 gpurun_out/r03_tests_adam.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03_bench_$name.json; echo "$name: $(cut -c1-230 gpurun_out/r03_bench_$name.json)"; }
b dcn_train --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
DIR_TRAIN_HIP_ADAM=0 b dcn_train_torch_adam --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
bash tools/prof.sh dcn_train -- --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_dcn_train.txt 2>&1; head -14 gpurun_out/prof_dcn_train.txt
