#!/bin/bash
# tools/prof_py.sh NAME -- python-args...   (GPU box): rocprofv3 kernel-trace summary of any python command -> gpurun_out/rNN_kernel_stats_NAME.csv
name=$1; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o $name -- python3 "$@" > gpurun_out/prof_$name.log 2>&1
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${ROUND:-r05}_kernel_stats_$name.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-100s %6s %12.1f us %6s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_$name
