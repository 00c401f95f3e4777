#!/bin/bash
# tools/r04_bucket_sweep.sh (GPU box): bucket_cap2_k variants under `bench.py --workload sharded_1gpu` -- kernel times from rocprofv3.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() {  # run TAG env...
  tag=$1; shift
  rm -rf gpurun_out/prof_bk_$tag
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bk_$tag -o $tag -- python3 bench.py --workload sharded_1gpu --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bk_$tag.log 2>&1 )
  f=$(find gpurun_out/prof_bk_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$tag" "$f" <<'PY'
import csv, sys
tag, f = sys.argv[1], sys.argv[2]
tot = 0.0
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("bucket_cap", "gather_slabs", "gather_onehot_k")):
        us = float(r["AverageNs"]) / 1e3
        tot += us
        out.append("%s %.1f" % (n.split("(")[0].replace("void dir::", "")[:28], us))
print("%-14s sum %.1f us | %s" % (tag, tot, " | ".join(out)))
PY
  rm -rf gpurun_out/prof_bk_$tag
}
run slabs_legacy DIR_SLABS_LEGACY=1
run slabs16 DIR_SLABS_LEGACY=0
