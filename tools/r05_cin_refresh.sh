#!/bin/bash
# tools/r05_cin_refresh.sh (GPU box): the CIN lines, traces and PMC summaries of profiles/r05_* again after the block restructure of cin_bf3_k
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r05_bench_$name.json; echo "$name: $(python3 -c "
import json; d=json.load(open('gpurun_out/r05_bench_$name.json')); print(round(d['ms_per_step'],4),'ms frac',round(d['roofline']['frac'],3))")"; }
b cin --workload cin --steps 5 --warmup 2
DIR_CIN_FWD_SPLIT=bf16x3 b cin_bf16x3 --workload cin --steps 5 --warmup 2 --no-cpu-baseline
b cin_backward --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline
DIR_CIN_BWD_SPLIT=bf16x3 DIR_DENSE_BWD_SPLIT=bf16x3 b cin_backward_bf16x3 --workload cin_backward --steps 5 --warmup 2 --no-cpu-baseline
b xdeepfm_full --workload xdeepfm_full --steps 10 --warmup 2 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 10 --warmup 2 --no-cpu-baseline
DIR_CIN_BWD_SPLIT=bf16x3 DIR_DENSE_BWD_SPLIT=bf16x3 b xdeepfm_train_bwd_bf16x3 --workload xdeepfm_train --steps 10 --warmup 2 --no-cpu-baseline
DIR_CIN_ROW_BITS_CARRY=0 b cin_rowscaled --workload cin --steps 5 --warmup 2 --no-cpu-baseline
DIR_CIN_FWD_SPLIT=f16x2_unscaled DIR_DENSE_BOUNDED_SPLIT=f16x2 DIR_DENSE_FWD_CARRY=0 b cin_r4_routing --workload cin --steps 5 --warmup 2 --no-cpu-baseline
for w in cin cin_backward xdeepfm_full xdeepfm_train; do
    ROUND=r05 DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 bash tools/prof.sh $w -- --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1; echo "== $w"; head -5 gpurun_out/prof_$w.txt | cut -c1-150
done
ROUND=r05 bash tools/pmc.sh cin_backward cin_ -- --workload cin_backward --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
ROUND=r05 bash tools/pmc.sh cin cin_ -- --workload cin --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
python3 bench.py --steps 50 --warmup 10 > gpurun_out/bench_default2.log 2>&1; grep '^{' gpurun_out/bench_default2.log | tail -1 > gpurun_out/r05_bench_default_head.json; python3 -c "
import json; d=json.load(open('gpurun_out/r05_bench_default_head.json')); print('default', round(d['ms_per_step'],4), d['roofline']['frac'], 'cfg5', d['secondary_cfg5_xdeepfm_cin']['ms_per_step'])"
