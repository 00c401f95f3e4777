#!/bin/bash
# tools/r06_round_final.sh (GPU box): the round-6 bench lines of every workload, kernel-trace summaries of the ones DESIGN.md quotes, the
# PMC traffic passes of the default command -> gpurun_out/r06_*
# usage: r06_round_final.sh [lines|traces|pmc|all]   (one gpurun call per part keeps each under the 20-minute limit)
part=${1:-all}
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r06_bench_$name.json; echo "$name: $(python3 -c "
import json,sys
d=json.load(open('gpurun_out/r06_bench_$name.json')); r=d['roofline']; print(round(d['ms_per_step'],4),'ms', 'frac', round(r['frac'],3), r.get('bound'))" 2>&1)"; }
# Warm-up lengths: the matrix-pipe / VALU-bound workloads warm up for >= 200 ms of back-to-back steps -- after the host-side set-up the GPU's clocks
# take tens of milliseconds of load to come up, and with round 5's (50, 5) the whole timed region lay inside that ramp (profiles/r06_warm.txt:
# mlp_dense 0.239 -> 0.204 ms, din 0.245 -> 0.201, cin 3.26 -> 2.97 on one box; the per-launch medians taken after the timed region never moved).
# HBM-bound lines keep round 5's settings: they read the same either way (the headline kernel a little FASTER cold).
if [ $part = lines ] || [ $part = all ]; then
b default --steps 200 --warmup 20
b gather_only --workload gather_only --steps 200 --warmup 20 --no-cpu-baseline
b fm_only --workload fm_only --steps 200 --warmup 20 --no-cpu-baseline
b linear --workload linear --steps 200 --warmup 20 --no-cpu-baseline
b deepfm_sparse_packed --workload deepfm_sparse_packed --steps 200 --warmup 20 --no-cpu-baseline
b multihot_bag --workload multihot_bag --steps 50 --warmup 5 --no-cpu-baseline
b dcn_cross --workload dcn_cross --steps 100 --warmup 10
b dcn_cross429 --workload dcn_cross --cross-d 429 --steps 100 --warmup 10 --no-cpu-baseline
b dcn_cross_backward --workload dcn_cross_backward --steps 100 --warmup 10 --no-cpu-baseline
b din --workload din --steps 200 --warmup 1000
DIR_DIN_PACKED=0 b din_wave --workload din --steps 200 --warmup 1000 --no-cpu-baseline
DIR_DIN_ARITH=f32 b din_f32 --workload din --steps 200 --warmup 400 --no-cpu-baseline
DIR_DIN_ARITH=bf16x3 b din_bf16x3 --workload din --steps 200 --warmup 600 --no-cpu-baseline
b din_full --workload din_full --steps 200 --warmup 800 --no-cpu-baseline
DIR_BENCH_DIN_ACT=dice b din_full_dice --workload din_full --steps 200 --warmup 600 --no-cpu-baseline
b din_train --workload din_train --steps 100 --warmup 100 --no-cpu-baseline
DIR_DIN_BWD_ARITH=f32 b din_train_bwd_f32 --workload din_train --steps 30 --warmup 60 --no-cpu-baseline
DIR_DIN_SAVE=0 b din_train_recompute --workload din_train --steps 30 --warmup 60 --no-cpu-baseline
b cin --workload cin --steps 10 --warmup 80
DIR_CIN_FWD_SPLIT=bf16x3 b cin_bf16x3 --workload cin --steps 10 --warmup 60 --no-cpu-baseline
DIR_CIN_POOLED_FUSED=0 b cin_two_pass_pooled --workload cin --steps 10 --warmup 80 --no-cpu-baseline
b cin_backward --workload cin_backward --steps 10 --warmup 30 --no-cpu-baseline
DIR_CIN_BWD_SPLIT=bf16x3 DIR_DENSE_BWD_SPLIT=bf16x3 b cin_backward_bf16x3 --workload cin_backward --steps 10 --warmup 25 --no-cpu-baseline
b mlp_dense --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_SPLIT=bf16x3 b mlp_dense_bf16x3 --workload mlp_dense --steps 200 --warmup 700 --no-cpu-baseline
DIR_BENCH_DENSE=layers b mlp_dense_layers --workload mlp_dense --steps 200 --warmup 800 --no-cpu-baseline
DIR_TOWER_KERNEL=rows b mlp_dense_rows --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_KERNEL=rows DIR_TOWER_RT=2 b mlp_dense_rt2 --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_RS=0 b mlp_dense_rs0 --workload mlp_dense --steps 200 --warmup 1000 --no-cpu-baseline
b deepfm_full --workload deepfm_full --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_RS=0 b deepfm_full_rs0 --workload deepfm_full --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_KERNEL=rows b deepfm_full_rows --workload deepfm_full --steps 200 --warmup 1000 --no-cpu-baseline
DIR_TOWER_SPLIT=bf16x3 b deepfm_full_bf16x3 --workload deepfm_full --steps 200 --warmup 700 --no-cpu-baseline
b dcn_full --workload dcn_full --steps 100 --warmup 400 --no-cpu-baseline
b esmm_full --workload esmm_full --steps 200 --warmup 600 --no-cpu-baseline
b xdeepfm_full --workload xdeepfm_full --steps 10 --warmup 70 --no-cpu-baseline
b deepfm_train --workload deepfm_train --steps 100 --warmup 150 --no-cpu-baseline
b deepfm_train_graph --workload deepfm_train --graph --steps 100 --warmup 150 --no-cpu-baseline
DIR_DENSE_BWD_SPLIT=bf16x3 b deepfm_train_bwd_bf16x3 --workload deepfm_train --steps 100 --warmup 150 --no-cpu-baseline
DIR_DENSE_BWD_SPLIT=bf16x3 b dcn_train_bwd_bf16x3 --workload dcn_train --steps 30 --warmup 40 --no-cpu-baseline
b esmm_train_graph --workload esmm_train --graph --steps 100 --warmup 100 --no-cpu-baseline
DIR_CIN_BWD_SPLIT=bf16x3 DIR_DENSE_BWD_SPLIT=bf16x3 b xdeepfm_train_bwd_bf16x3 --workload xdeepfm_train --steps 10 --warmup 15 --no-cpu-baseline
b dcn_train --workload dcn_train --steps 30 --warmup 40 --no-cpu-baseline
b esmm_train --workload esmm_train --steps 100 --warmup 100 --no-cpu-baseline
b xdeepfm_train --workload xdeepfm_train --steps 10 --warmup 20 --no-cpu-baseline
b train_sparse --workload train_sparse --steps 100 --warmup 10 --no-cpu-baseline
b train_sparse_graph --workload train_sparse --graph --steps 100 --warmup 10 --no-cpu-baseline
b sharded_1gpu --workload sharded_1gpu --steps 100 --warmup 10 --no-cpu-baseline
b sharded_deepfm_1gpu --workload sharded_deepfm_1gpu --steps 100 --warmup 10 --no-cpu-baseline
DIR_BENCH_SHARD_CONSUME=0 b sharded_deepfm_1gpu_finish --workload sharded_deepfm_1gpu --steps 100 --warmup 10 --no-cpu-baseline
b transform --workload transform --steps 100 --warmup 10 --no-cpu-baseline
b small_batch --workload small_batch --steps 200 --warmup 20 --no-cpu-baseline
DIR_BENCH_SMALL_BATCH=100 b small_batch_100 --workload small_batch --steps 200 --warmup 20 --no-cpu-baseline
DIR_BENCH_SMALL_BATCH=1024 b small_batch_1024 --workload small_batch --steps 200 --warmup 20 --no-cpu-baseline
DIR_BENCH_SMALL_BATCH=2048 b small_batch_2048 --workload small_batch --steps 200 --warmup 20 --no-cpu-baseline
DIR_BENCH_SMALL_BATCH=4096 b small_batch_4096 --workload small_batch --steps 200 --warmup 20 --no-cpu-baseline
DIR_BENCH_DIN_ACT=dice b din_train_dice --workload din_train --steps 20 --warmup 30 --no-cpu-baseline
DIR_BENCH_DIN_ACT=prelu b din_train_prelu --workload din_train --steps 20 --warmup 40 --no-cpu-baseline
DIR_BENCH_DIN_ACT=dice DIR_DICE_FUSED_BWD=0 b din_train_dice_3k --workload din_train --steps 20 --warmup 30 --no-cpu-baseline
DIR_BENCH_DIN_ACT=dice DIR_DIN_ROWS_TRAIN=0 b din_train_dice_torch --workload din_train --steps 5 --warmup 2 --no-cpu-baseline
DIR_CIN_ROW_BITS_CARRY=0 b cin_rowscaled --workload cin --steps 10 --warmup 80 --no-cpu-baseline
fi
if [ $part = traces ] || [ $part = all ]; then
for w in default deepfm_full esmm_full dcn_full dcn_train deepfm_train esmm_train train_sparse sharded_1gpu cin cin_backward multihot_bag din din_full din_train xdeepfm_full xdeepfm_train small_batch; do
    if [ $w = default ]; then a="--steps 100 --warmup 10 --no-cpu-baseline"; elif [ $w = din ] || [ $w = din_full ]; then a="--workload $w --steps 200 --warmup 800 --no-cpu-baseline"; elif [ $w = din_train ]; then a="--workload $w --steps 50 --warmup 100 --no-cpu-baseline"; elif [ $w = deepfm_full ] || [ $w = esmm_full ] || [ $w = dcn_full ] || [ $w = small_batch ] || [ $w = train_sparse ] || [ $w = sharded_1gpu ] || [ $w = multihot_bag ]; then a="--workload $w --steps 100 --warmup 400 --no-cpu-baseline"; else a="--workload $w --steps 10 --warmup 40 --no-cpu-baseline"; fi
    ROUND=r06 DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 bash tools/prof.sh $w -- $a > gpurun_out/prof_$w.txt 2>&1; echo "== $w"; head -5 gpurun_out/prof_$w.txt | cut -c1-150
done
# library GEMM kernels (Tensile: "Cijk_...") in any of the traced steps; tools/libcall_probe.py is the Python-side view of the same question
{ echo "traced workloads whose kernel list holds a Tensile (rocBLAS / hipBLASLt) GEMM:"; grep -l "Cijk_" gpurun_out/r06_kernel_stats_*.csv || echo "  none"; } > gpurun_out/r06_library_kernels.txt; cat gpurun_out/r06_library_kernels.txt
ROUND=r06 DIR_BENCH_DIN_ACT=dice bash tools/prof.sh din_train_dice -- --workload din_train --steps 10 --warmup 30 --no-cpu-baseline > gpurun_out/prof_din_train_dice.txt 2>&1; head -5 gpurun_out/prof_din_train_dice.txt | cut -c1-150
ROUND=r06 bash tools/traffic.sh > gpurun_out/traffic_r06.txt 2>&1; tail -16 gpurun_out/traffic_r06.txt
fi
if [ $part = pmc ] || [ $part = all ]; then
export ROUND=r06
bash tools/pmc.sh din din_pack_k -- --workload din --steps 5 --warmup 1 --no-cpu-baseline
DIR_DIN_PACKED=0 bash tools/pmc.sh din_wave din_wave_k -- --workload din --steps 5 --warmup 1 --no-cpu-baseline
bash tools/pmc.sh tower tower_cs_k -- --workload mlp_dense --steps 5 --warmup 1 --no-cpu-baseline
DIR_TOWER_KERNEL=rows bash tools/pmc.sh tower_rows tower_bf3_k -- --workload mlp_dense --steps 5 --warmup 1 --no-cpu-baseline
bash tools/pmc.sh cin cin_ -- --workload cin --steps 3 --warmup 1 --no-cpu-baseline
bash tools/pmc.sh cin_backward cin_ -- --workload cin_backward --steps 3 --warmup 1 --no-cpu-baseline
bash tools/pmc_kernels.sh sharded_1gpu
fi
