cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_shard_multiproc.py -x -q -m gpu -k "shard or bucket or capacity or two_ranks or trainer" > gpurun_out/r03_tests_shard.log 2>&1; echo "shard tests rc=$?"; tail -5 gpurun_out/r03_tests_shard.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03_bench_$name.json; echo "$name: $(cut -c1-300 gpurun_out/r03_bench_$name.json)"; }
DIR_BENCH_CFG5_SHARDED=1 b default_cfg5_sharded --steps 50 --warmup 10 --no-cpu-baseline
b default --steps 200 --warmup 20
b dcn_cross --workload dcn_cross --steps 100 --warmup 10
b din --workload din --steps 50 --warmup 5
b cin --workload cin --steps 5 --warmup 2
