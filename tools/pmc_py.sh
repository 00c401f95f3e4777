#!/bin/bash
# tools/pmc_py.sh KERNEL_SUBSTRING "COUNTERS" script.py [args]  : per-launch mean of PMC counters for kernels matching, for any python script
sub=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_tmp
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_tmp -o t -- python3 "$@" > gpurun_out/pmc_tmp.log 2>&1
f=$(find gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
python3 - "$f" "$sub" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, len(disp[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.4g" % (c, v / len(disp[k])))
PY
rm -rf gpurun_out/pmc_tmp
