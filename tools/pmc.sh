#!/bin/bash
# tools/pmc.sh NAME KERNEL_SUBSTRING -- bench.py args...   (GPU box)
# One rocprofv3 --pmc pass (counters only: never combined with tracing) of a bench.py command; per-launch means of the SQ counters of
# the kernels whose name contains KERNEL_SUBSTRING -> gpurun_out/r02_pmc_NAME.json (MFMA-pipe busy fraction, wait fractions).
name=$1; sub=$2; shift; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$name
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d gpurun_out/pmc_$name -o $name -- python3 bench.py "$@" > gpurun_out/pmc_$name.log 2>&1
f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
python3 - "$f" "$sub" "$name" "$*" <<'PY'
import csv, json, sys, collections
f, sub, name, cmd = sys.argv[1:5]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
out = {"command": "rocprofv3 --pmc <SQ group + GRBM_GUI_ACTIVE> --output-format csv -- python3 bench.py " + cmd, "kernels": {}}
for k, c in acc.items():
    n = len(disp[k])
    m = {kk: v / n for kk, v in c.items()}
    d = {}
    if m.get("GRBM_GUI_ACTIVE"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
        d["gpu_cycles_per_launch"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            d["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)   # 1024 SIMDs
    if m.get("SQ_WAVE_CYCLES"):
        d["wave_wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
        d["wave_wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
        d["wave_active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
    out["kernels"][k[:120]] = {"launches": n, "per_launch_mean": m, "derived": d}
json.dump(out, open("gpurun_out/%s_pmc_%s.json" % (__import__("os").environ.get("ROUND","r03"), name), "w"), indent=1)
for k, v in out["kernels"].items():
    print(k[:80], v["launches"], json.dumps(v["derived"]))
PY
rm -rf gpurun_out/pmc_$name
