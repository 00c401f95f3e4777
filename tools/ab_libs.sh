#!/bin/bash
# tools/ab_libs.sh "LIB_A LIB_B ..." "workload[:bench args] ..."   (GPU box): the same bench.py lines under different builds of the library
# (DIR_HIP_LIBRARY), alternating A B A B on ONE box -> ms_per_step per build and workload.  Libraries are paths relative to the repo root.
cd "$GRAFT_REPO_ROOT"
libs=($1); shift
for spec in $1; do
    w=${spec%%:*}; extra=""; [ "$spec" != "$w" ] && extra=${spec#*:}
    line="$w"
    for rep in 1 2; do
        for lib in "${libs[@]}"; do
            ms=$(DIR_HIP_LIBRARY=$PWD/$lib DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 timeout -k 10 300 python3 bench.py --workload $w --steps ${STEPS:-30} --warmup ${WARMUP:-5} --no-cpu-baseline ${extra//,/ } 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
            line="$line  $(basename $lib .so)=$ms"
        done
    done
    echo "$line"
done
