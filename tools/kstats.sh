#!/bin/bash
# tools/kstats.sh WORKLOAD [bench args...]  (GPU box): rocprofv3 --kernel-trace --stats of one bench line -> the top kernels by total time
# (gpurun_out/kstats_WORKLOAD_kernel_stats.csv is the full table).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
w=$1; shift
rm -rf gpurun_out/kstats_tmp
DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_tmp -o t -- python3 bench.py --workload $w --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline "$@" > gpurun_out/kstats_$w.log 2>&1
f=$(find gpurun_out/kstats_tmp -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/kstats_${w}_kernel_stats.csv && rm -rf gpurun_out/kstats_tmp
python3 - gpurun_out/kstats_${w}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(__import__("os").environ.get("TOP", "18"))]:
    print("%-84s %5s %9.1f us %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
