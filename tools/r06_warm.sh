#!/bin/bash
# tools/r06_warm.sh (GPU box): bench lines at several (steps, warmup): how much of a short timed region is the clock ramp after idle?
cd "$GRAFT_REPO_ROOT"
for wl in ${WLS:-deepfm_gather_fm mlp_dense deepfm_full}; do
for cfg in "20 5" "50 5" "200 50" "1000 200"; do set -- $cfg
  DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 timeout -k 10 300 python3 bench.py --workload $wl --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl steps $1 warmup $2: ms_per_step %.4f  frac %.3f  median %.1f p10 %.1f p90 %.1f' % (d['ms_per_step'], r['frac'], r.get('launch_us_median',0), r.get('launch_us_p10',0), r.get('launch_us_p90',0)))" || exit 1
done; done
