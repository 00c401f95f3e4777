#!/bin/bash
# tools/ab_env_multi.sh "ENV1=a ENV2=b|ENV1=c|..." WORKLOAD [bench args]  (GPU box): one bench line under several environments, two rounds, on
# ONE box -> ms_per_step and the per-launch p10 / median / p90 of each
cd "$GRAFT_REPO_ROOT"
IFS='|' read -ra envs <<< "$1"; shift
w=$1; shift
for rep in 1 2; do
  for e in "${envs[@]}"; do
    echo "[$e] $w: $(env $e python3 bench.py --workload $w "$@" --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4),'ms  p10/med/p90', round(r.get('launch_us_p10',0)), round(r.get('launch_us_median',0)), round(r.get('launch_us_p90',0)))")"
  done
done
