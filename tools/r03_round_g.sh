#!/bin/bash
# round 3, step g: DIN profiles at HEAD, din_train with the static schedule in the backward's row pass as well
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03g_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03g_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
b din_train --workload din_train --steps 30 --warmup 5 --no-cpu-baseline
DIR_DIN_STATIC=1 b din_train_static --workload din_train --steps 30 --warmup 5 --no-cpu-baseline
for w in din din_train; do
    DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh $w -- --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1; echo "== $w"; head -8 gpurun_out/prof_$w.txt | cut -c1-150
done
DIR_DIN_STATIC=1 DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh din_train_static -- --workload din_train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_din_train_static.txt 2>&1; echo "== din_train static"; head -6 gpurun_out/prof_din_train_static.txt | cut -c1-150
