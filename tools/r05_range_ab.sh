#!/bin/bash
# tools/r05_range_ab.sh (GPU box): what the scale-free fp16 x 2 forward paths of round 5 cost against round 4's unscaled routing, same box.
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), 'ms')"; }
for w in "cin --steps 5 --warmup 2" "xdeepfm_full --steps 10 --warmup 2" "deepfm_train --steps 100 --warmup 10" "dcn_full --steps 50 --warmup 5" "esmm_full --steps 50 --warmup 5" "esmm_train --steps 100 --warmup 10" "esmm_train --graph --steps 100 --warmup 10" "dcn_train --steps 30 --warmup 5" "mlp_dense --steps 50 --warmup 5"; do
  set -- $w; n=$1; shift
  b "$n[r5] $*" --workload $n "$@"
  DIR_DENSE_BOUNDED_SPLIT=f16x2 DIR_CIN_FWD_SPLIT=f16x2_unscaled DIR_DENSE_FWD_CARRY=0 b "$n[r4-routing] $*" --workload $n "$@"
done
