#!/bin/bash
# tools/r04_refresh_train.sh (GPU box): the training-step lines and traces only (after a change that touches the dense backward)
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r04_bench_$name.json; echo "$name: $(python3 -c "
import json
print(round(json.load(open('gpurun_out/r04_bench_$name.json'))['ms_per_step'],4))")"; }
b deepfm_train --workload deepfm_train --steps 100 --warmup 10 --no-cpu-baseline
b deepfm_train_graph --workload deepfm_train --graph --steps 100 --warmup 10 --no-cpu-baseline
b esmm_train --workload esmm_train --steps 100 --warmup 10 --no-cpu-baseline
b esmm_train_graph --workload esmm_train --graph --steps 100 --warmup 10 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 30 --warmup 5 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 10 --warmup 2 --no-cpu-baseline
for w in deepfm_train esmm_train dcn_train xdeepfm_train; do
    ROUND=r04 DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 bash tools/prof.sh $w -- --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1
done
{ echo "traced workloads whose kernel list holds a Tensile (rocBLAS / hipBLASLt) GEMM:"; grep -l "Cijk_" gpurun_out/r04_kernel_stats_*.csv || echo "  none"; } > gpurun_out/r04_library_kernels.txt; cat gpurun_out/r04_library_kernels.txt
