#!/bin/bash
# tools/r02_final_round.sh (GPU box): end-of-round-2 refresh of the bench lines, kernel-trace summaries and PMC passes that changed with the
# CIN backward work (two-field data-gradient chunks, pipelined weight-gradient loads) -> gpurun_out/
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r02_bench_$name.json; echo "$name: $(cut -c1-230 gpurun_out/r02_bench_$name.json)"; }
b cin_backward --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 5 --warmup 2 --no-cpu-baseline
b xdeepfm_full --workload xdeepfm_full --steps 10 --warmup 3 --no-cpu-baseline
b cin --workload cin --steps 10 --warmup 3 --no-cpu-baseline
b din --workload din --steps 50 --warmup 5 --no-cpu-baseline
b din_train --workload din_train --steps 20 --warmup 3 --no-cpu-baseline
b deepfm_train --workload deepfm_train --steps 20 --warmup 3 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
b esmm_train --workload esmm_train --steps 20 --warmup 3 --no-cpu-baseline
python3 tools/sweep_shapes.py > gpurun_out/r02_sweep_shapes.md 2> gpurun_out/sweep.err; tail -3 gpurun_out/sweep.err
python3 tools/cin_bf3_stress.py > gpurun_out/cin_bf3_stress.log 2>&1; tail -4 gpurun_out/cin_bf3_stress.log
bash tools/prof.sh cin_backward -- --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_cin_backward.txt 2>&1
bash tools/prof.sh deepfm_train -- --workload deepfm_train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_deepfm_train.txt 2>&1
bash tools/prof.sh dcn_train -- --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_dcn_train.txt 2>&1
bash tools/prof.sh esmm_train -- --workload esmm_train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_esmm_train.txt 2>&1
bash tools/pmc.sh cin_backward cin_ -- --workload cin_backward --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_cin_backward.txt 2>&1
head -6 gpurun_out/prof_cin_backward.txt; cat gpurun_out/pmc_cin_backward.txt | cut -c1-260
timeout -k 10 60 tools/valu_rate_probe > gpurun_out/r02_valu_rate_probe.txt 2>&1; cat gpurun_out/r02_valu_rate_probe.txt
