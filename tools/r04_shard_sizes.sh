#!/bin/bash
# tools/r04_shard_sizes.sh (GPU box): the sharded lookup's rank-local kernels at the table size ONE rank holds: the whole 26 x 1 M rows
# (what `sharded_1gpu` measures on one GPU) and 26 x 125 000 rows (a rank's 'div' shard at P = 8: 208 MB, inside the Infinity Cache).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in 1000000 125000; do
  rm -rf gpurun_out/prof_ss_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ss_$v -o s -- python3 bench.py --workload sharded_1gpu --vocab $v --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_ss_$v.log 2>&1
  f=$(find gpurun_out/prof_ss_$v -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/r04_kernel_stats_sharded_1gpu_vocab$v.csv
  python3 - "$v" "$f" <<'PY'
import csv, sys
v, f = sys.argv[1], sys.argv[2]
tot, out = 0.0, []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("bucket_cap", "gather_slabs", "gather_onehot_k")):
        us = float(r["AverageNs"]) / 1e3
        tot += us
        out.append("%s %.1f" % (n.split("(")[0].replace("void dir::", "")[:26], us))
print("vocab/field %-8s sum %.1f us | %s" % (v, tot, " | ".join(out)))
PY
  rm -rf gpurun_out/prof_ss_$v
done
