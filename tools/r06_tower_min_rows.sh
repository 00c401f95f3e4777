#!/bin/bash
# tools/r06_tower_min_rows.sh (GPU box): DeepFM / DCN forward latency from a HIP graph at B = 256 .. 4096 with the one-launch tower admitted from
# DIR_TOWER_MIN_ROWS rows on (default 4096): is the 64-row-tile kernel (tower_cs_k) the better route for mid-size batches?
cd "$GRAFT_REPO_ROOT"
for B in 256 512 1024 2048 4096; do for mr in 4096 64; do
  DIR_TOWER_MIN_ROWS=$mr DIR_BENCH_SMALL_BATCH=$B timeout -k 10 300 python3 bench.py --workload small_batch --steps 200 --warmup 200 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['config']['latency_per_forward']
print('B $B min_rows $mr: deepfm graph %.1f us eager %.1f   dcn graph %.1f us' % (l['deepfm']['graph_replay_us'], l['deepfm']['eager_us'], l['dcn']['graph_replay_us']))" || exit 1
done; done
