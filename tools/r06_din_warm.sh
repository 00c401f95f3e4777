#!/bin/bash
# tools/r06_din_warm.sh (GPU box): the din line at two warm-up lengths -- is the 50-step average (0.243 ms) against the 200-launch median (0.214) a clock ramp?
cd "$GRAFT_REPO_ROOT"
for cfg in "50 5" "200 50" "400 100" "50 5"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --workload din --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('steps $1 warmup $2: ms_per_step %.4f  median %.1f p10 %.1f p90 %.1f  hbm_frac %.3f' % (d['ms_per_step'], r['launch_us_median'], r['launch_us_p10'], r['launch_us_p90'], r['hbm_frac']))" || exit 1
done
