#!/bin/bash
# tools/pmc2.sh NAME KERNEL_SUBSTRING "COUNTER LIST" -- bench.py args...  : like pmc.sh with a caller-chosen counter list (<= 8 SQ counters)
name=$1; sub=$2; ctrs=$3; shift; shift; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$name
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$name -o $name -- python3 bench.py "$@" > gpurun_out/pmc_$name.log 2>&1
f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
python3 - "$f" "$sub" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
for k, v in sorted(acc.items()):
    print("%-28s %.4g per launch" % (k, v / max(1, len(disp))))
PY
rm -rf gpurun_out/pmc_$name
