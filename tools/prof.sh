#!/bin/bash
# tools/prof.sh NAME -- bench.py args...   (GPU box): rocprofv3 kernel-trace summary of one bench.py command -> gpurun_out/prof_NAME/,
# top kernels printed, the stats CSV copied to gpurun_out/rNN_kernel_stats_NAME.csv for profiles/.
name=$1; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o $name -- python3 bench.py "$@" > gpurun_out/prof_$name.log 2>&1
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${ROUND:-r03}_kernel_stats_$name.csv
grep '^{' gpurun_out/prof_$name.log | tail -1 > gpurun_out/${ROUND:-r03}_profiled_bench_$name.json
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-100s %6s %12.1f us %6s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_$name   # the trace itself is large; the summary is what is kept
