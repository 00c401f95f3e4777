#!/bin/bash
# tools/r03_round_c.sh (GPU box): the fused tower kernel: tests, bench lines (tower vs per-layer), kernel trace; the cfg-5 overlap with
# CU-masked side streams
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_tower.py -x -q -m gpu > gpurun_out/r03_tests_tower.log 2>&1; echo "tower tests rc=$?"; tail -6 gpurun_out/r03_tests_tower.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "dedup or bucket" > gpurun_out/r03_tests_dedup.log 2>&1; echo "dedup tests rc=$?"; tail -3 gpurun_out/r03_tests_dedup.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03_bench_$name.json; echo "$name: $(cut -c1-330 gpurun_out/r03_bench_$name.json)"; }
b mlp_dense --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
DIR_BENCH_DENSE=layers b mlp_dense_layers --workload mlp_dense --steps 50 --warmup 5 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 50 --warmup 5 --no-cpu-baseline
for cus in 0 8 16 32; do DIR_BENCH_CFG5_SIDE_CUS=$cus DIR_BENCH_CFG5_SHARDED=1 b cfg5_side$cus --steps 20 --warmup 5 --no-cpu-baseline; python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_cfg5_side$cus.json'))['secondary_cfg5_xdeepfm_cin']; print('  side_cus=$cus', d.get('ms_per_step'), d.get('cin_only_ms_per_step'), d.get('lookup_exposed_frac'), d.get('error'))"; done
bash tools/prof.sh deepfm_full -- --workload deepfm_full --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_deepfm_full.txt 2>&1; head -8 gpurun_out/prof_deepfm_full.txt
