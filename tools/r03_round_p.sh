#!/bin/bash
# round 3, step p: row pass + weight-gradient pass chunk by chunk (records stay in the Infinity Cache between the two)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_backward.py -q -x -k "din" > gpurun_out/r03_p_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_p_tests.log
for c in 0 4096 8192 16384 0 8192; do
  DIR_DIN_BWD_CHUNK=$c timeout -k 10 300 python3 bench.py --workload din_train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_din_train_c$c.log 2>&1; echo "din_train chunk $c: $(grep '^{' gpurun_out/bench_din_train_c$c.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
done
DIR_BENCH_NO_SECONDARY=1 bash tools/prof.sh din_train -- --workload din_train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_din_train.txt 2>&1; head -4 gpurun_out/prof_din_train.txt | cut -c1-140
