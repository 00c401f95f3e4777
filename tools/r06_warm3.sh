#!/bin/bash
# tools/r06_warm3.sh (GPU box): the driver's command (--steps 20 --warmup 5) five times per setting: settle phase on / off, polled closing event on / off
cd "$GRAFT_REPO_ROOT"
for cfg in "250 1" "0 1" "250 0" "0 0" "60 1"; do set -- $cfg
for rep in 1 2 3 4 5; do
  DIR_BENCH_SETTLE_MS=$1 DIR_BENCH_SPIN=$2 DIR_BENCH_NO_SWEEP=1 DIR_BENCH_NO_SECONDARY=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('settle $1 spin $2: ms_per_step %.4f  frac %.3f  hip_events %.1f us median %.1f' % (d['ms_per_step'], r['frac'], r['hip_event_avg_launch_us'], r.get('launch_us_median',0)))" || exit 1
done; done
