cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_tower.py -x -q -m gpu > gpurun_out/r03_tests_tower.log 2>&1; echo "tower tests rc=$?"; tail -6 gpurun_out/r03_tests_tower.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03_bench_$name.json; echo "$name: $(cut -c1-230 gpurun_out/r03_bench_$name.json)"; }
b mlp_dense --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline
