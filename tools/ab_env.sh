#!/bin/bash
# tools/ab_env.sh VAR WORKLOAD [steps] (GPU box): ms_per_step of a bench workload with VAR=0 / VAR=1, twice each, interleaved
cd "$GRAFT_REPO_ROOT"
var=$1; w=$2; s=${3:-100}
for v in 0 1 0 1; do
    echo "$var=$v $w: $(env $var=$v python3 bench.py --workload $w --steps $s --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))") ms"
done
