#!/bin/bash
# tools/r04_sort_ab.sh (GPU box): the slot-major sort of one-hot entries against the global-key sort (DIR_SORT=global) on the workloads that sort
cd "$GRAFT_REPO_ROOT"
ms() { python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for w in train_sparse deepfm_train esmm_train dcn_train; do
    a=$(python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | ms)
    b=$(DIR_SORT=global python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | ms)
    echo "$w: slot-major $a ms, global keys $b ms"
done
