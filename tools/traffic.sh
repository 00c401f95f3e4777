#!/bin/bash
# tools/traffic.sh (GPU box): HBM-side bytes per launch of the default bench's gather kernel from rocprofv3 PMC, as
# MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (never combined with tracing), values in KB,
# gfx950 correction: a wide coalesced streaming read (the int64 id stream) is tallied at half and is doubled; single 64-byte row requests
# are counted exactly (FETCH_SIZE*1024 = TCC_EA0_RDREQ*64).  -> gpurun_out/${ROUND:-r04}_pmc_traffic.json + gpurun_out/traffic.json (copy to profiles/)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for pass in "FETCH_SIZE TCC_EA0_RDREQ_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_t_$tag
  DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_t_$tag -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/pmc_t_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
B, F, K = 65536, 26, 16
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_t_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gather_onehot_k" in k and "true, true, true" in k.replace("(bool)1", "true"):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
m = {c: acc[c] / len(n[c]) for c in acc}
rows = B * F * 4 * K
fetch = m["FETCH_SIZE"] * 1024
ids_half = fetch - rows                      # what is left after the exactly-counted 64-byte row requests: the id stream at half
read = rows + 2 * ids_half
write = m["WRITE_SIZE"] * 1024
out = {"command": "DIR_BENCH_NO_SECONDARY=1 DIR_BENCH_NO_SWEEP=1 rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline",
       "kernel": "gather_onehot_k<4,4,16,26,fm,out,nt> (uniform ids)", "launches_averaged": {c: len(n[c]) for c in n},
       "per_launch_mean": m,
       "derived": {"FETCH_bytes": fetch, "RDREQ_x64B": m.get("TCC_EA0_RDREQ_sum", 0) * 64, "row_read_bytes_exact": rows,
                   "id_stream_bytes_counted_at_half": ids_half, "read_bytes_corrected": read, "WRITE_bytes": write,
                   "WRREQ_x64B": m.get("TCC_EA0_WRREQ_sum", 0) * 64, "traffic_bytes_per_launch": read + write,
                   "algorithmic_bytes_per_launch": B * (F * (8 + 2 * 4 * K) + 4),
                   "L2_hit_rate": m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))}}
out["derived"]["traffic_over_algorithmic"] = out["derived"]["traffic_bytes_per_launch"] / out["derived"]["algorithmic_bytes_per_launch"]
import hashlib, os, time
src = "details-in-recommendation_amd/csrc/embedding_bag.hip"
stamp = "%s, %s sha256 %s" % (time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()), src, hashlib.sha256(open(src, "rb").read()).hexdigest()[:16])
out["measured_at"] = stamp
rnd = os.environ.get("ROUND", "r04")
json.dump(out, open("gpurun_out/%s_pmc_traffic.json" % rnd, "w"), indent=1)
json.dump({"_comment": "HBM-side bytes per launch of gather_onehot_k<fm,out> (uniform ids, non-temporal row loads) from rocprofv3 PMC, separate passes "
                       "for FETCH_SIZE and WRITE_SIZE (tools/traffic.sh -> profiles/%s_pmc_traffic.json): row reads are single 64-byte requests and "
                       "counted exactly, the int64 id stream is a wide coalesced read that gfx950 tallies at half and is doubled "
                       "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  The counters sit at the L2<->fabric boundary." % rnd,
           "_measured_at": stamp, "deepfm_gather_fm": int(round(out["derived"]["traffic_bytes_per_launch"]))},
          open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
PY
rm -rf gpurun_out/pmc_t_*
