#!/bin/bash
# tools/r04_units1_ab.sh (GPU box): the training steps after the units = 1 forward kernel and the MFMA weight-gradient kernel for narrow layers
cd "$GRAFT_REPO_ROOT"
ms() { python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for w in esmm_train deepfm_train dcn_train din_train xdeepfm_train; do
    s=100; [ $w = xdeepfm_train ] && s=10; [ $w = dcn_train ] && s=30
    a=$(python3 bench.py --workload $w --steps $s --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | ms)
    echo "$w: $a ms"
done
a=$(python3 bench.py --workload esmm_train --graph --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | ms); echo "esmm_train --graph: $a ms"
