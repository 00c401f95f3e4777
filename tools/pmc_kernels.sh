#!/bin/bash
# tools/pmc_kernels.sh WORKLOAD [bench args...] (GPU box): HBM-side counters PER KERNEL of one bench workload from rocprofv3 PMC -- FETCH_SIZE /
# WRITE_SIZE (KB) with the request counts, L2 hits / misses -- in separate passes, never combined with tracing (MI355X_MICROARCH.md, HBM
# section).  Raw means per launch; reads that are wide coalesced streams are tallied at half by gfx950's FETCH_SIZE (RDREQ x 64 B is the
# exact figure for 64-byte requests).  -> gpurun_out/${ROUND:-r04}_pmc_kernels_WORKLOAD.json
w=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for pass in "FETCH_SIZE TCC_EA0_RDREQ_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_k_$tag
  DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_k_$tag -o t -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/pmc_k_$tag.log 2>&1
done
python3 - "$w" <<'PY'
import csv, glob, json, collections, os, sys, re
w = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmc_k_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "dir::" not in k:
            continue
        k = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "")).replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {"command": "DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --workload %s --steps 10 --warmup 2 --no-cpu-baseline" % w,
       "note": "means per launch; FETCH_SIZE / WRITE_SIZE in KB as the counters report them; *_MB derived (x 1024 / 1e6); RDREQ / WRREQ x 64 B", "kernels": {}}
for k in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", 0) - acc[k].get("WRITE_SIZE", 0)):
    m = {c: acc[k][c] / len(n[k][c]) for c in acc[k]}
    d = {"launches": max(len(v) for v in n[k].values()), "per_launch_mean": m,
         "fetch_MB": m.get("FETCH_SIZE", 0) * 1024 / 1e6, "write_MB": m.get("WRITE_SIZE", 0) * 1024 / 1e6,
         "rdreq_x64B_MB": m.get("TCC_EA0_RDREQ_sum", 0) * 64 / 1e6, "wrreq_x64B_MB": m.get("TCC_EA0_WRREQ_sum", 0) * 64 / 1e6,
         "L2_hit_rate": m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))}
    out["kernels"][k] = d
    print("%-60s fetch %8.1f MB  write %8.1f MB  L2 hit %.2f" % (k[:60], d["fetch_MB"], d["write_MB"], d["L2_hit_rate"]))
json.dump(out, open("gpurun_out/%s_pmc_kernels_%s.json" % (os.environ.get("ROUND", "r04"), w), "w"), indent=1)
PY
rm -rf gpurun_out/pmc_k_FETCH_SIZE gpurun_out/pmc_k_WRITE_SIZE gpurun_out/pmc_k_TCC_HIT_sum
