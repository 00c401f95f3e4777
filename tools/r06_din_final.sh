#!/bin/bash
# tools/r06_din_final.sh (GPU box): the DIN lines, trace and counters of tools/r06_round_final.sh alone (after the mixed-precision split, NOTES R6.11)
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r06_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r06_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), 'hbm_frac', round(r.get('hbm_frac',0),3), round(r.get('hbm_frac_length_aware',0),3))" 2>&1)"; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_din_pack.py tests/test_gpu_parity.py -q -k "din" 2>&1 | tail -2
b din --workload din --steps 200 --warmup 1000
DIR_DIN_PACKED=0 b din_wave --workload din --steps 200 --warmup 1000 --no-cpu-baseline
b din_full --workload din_full --steps 200 --warmup 800 --no-cpu-baseline
DIR_BENCH_DIN_ACT=dice b din_full_dice --workload din_full --steps 200 --warmup 600 --no-cpu-baseline
ROUND=r06 DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 bash tools/prof.sh din -- --workload din --steps 200 --warmup 800 --no-cpu-baseline > gpurun_out/prof_din.txt 2>&1; head -3 gpurun_out/prof_din.txt | cut -c1-150
ROUND=r06 DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 bash tools/prof.sh din_full -- --workload din_full --steps 200 --warmup 800 --no-cpu-baseline > gpurun_out/prof_din_full.txt 2>&1; head -3 gpurun_out/prof_din_full.txt | cut -c1-150
export ROUND=r06
bash tools/pmc.sh din din_pack_k -- --workload din --steps 5 --warmup 1 --no-cpu-baseline
