cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r05_bench_$name.json; echo "$name: $(python3 -c "
import json; d=json.load(open('gpurun_out/r05_bench_$name.json')); print(round(d['ms_per_step'],4),'ms frac',round(d['roofline']['frac'],3))")"; }
b deepfm_sparse_packed --workload deepfm_sparse_packed --steps 200 --warmup 20 --no-cpu-baseline
b sharded_1gpu --workload sharded_1gpu --steps 100 --warmup 10 --no-cpu-baseline
b sharded_deepfm_1gpu --workload sharded_deepfm_1gpu --steps 100 --warmup 10 --no-cpu-baseline
DIR_BENCH_SHARD_CONSUME=0 b sharded_deepfm_1gpu_finish --workload sharded_deepfm_1gpu --steps 100 --warmup 10 --no-cpu-baseline
b mlp_dense --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
ROUND=r05 DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 bash tools/prof.sh sharded_deepfm_1gpu -- --workload sharded_deepfm_1gpu --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof_sd.txt 2>&1; head -6 gpurun_out/prof_sd.txt | cut -c1-150
