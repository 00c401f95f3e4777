#!/bin/bash
# tools/r02_final_round.sh (GPU box): end-of-round-2 refresh of the bench lines and kernel-trace summaries of the training workloads
# (CIN backward regrouping, dense weight / bias gradient kernels, fused head backward) and the probes quoted in DESIGN.md -> gpurun_out/
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r02_bench_$name.json; echo "$name: $(cut -c1-200 gpurun_out/r02_bench_$name.json)"; }
b default --steps 200 --warmup 20
b cin_backward --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 10 --warmup 2 --no-cpu-baseline
b din --workload din --steps 50 --warmup 5 --no-cpu-baseline
b din_train --workload din_train --steps 30 --warmup 5 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline
b deepfm_train --workload deepfm_train --steps 30 --warmup 5 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 10 --warmup 2 --no-cpu-baseline
b esmm_train --workload esmm_train --steps 30 --warmup 5 --no-cpu-baseline
for w in cin_backward deepfm_train dcn_train esmm_train din_train; do
    bash tools/prof.sh $w -- --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1
done
head -6 gpurun_out/prof_deepfm_train.txt gpurun_out/prof_esmm_train.txt
python3 tools/dense_dw_probe.py > gpurun_out/r02_dense_dw_probe.txt 2>&1; tail -11 gpurun_out/r02_dense_dw_probe.txt
