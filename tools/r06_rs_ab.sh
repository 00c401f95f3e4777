#!/bin/bash
# tools/r06_rs_ab.sh (GPU box): the tower tests, then mlp_dense / deepfm_full with the row scaling on and off (DIR_TOWER_RS), twice, one box
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_tower.py -x -q 2>&1 | tail -3 || exit 1
for rs in 1 0 1 0; do
  for w in mlp_dense deepfm_full; do
    DIR_TOWER_RS=$rs timeout -k 10 200 python3 bench.py --workload $w --steps 200 --warmup 1000 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rs=$rs $w', round(d['ms_per_step'],4))" || exit 1
  done
done
