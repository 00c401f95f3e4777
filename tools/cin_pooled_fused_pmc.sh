#!/bin/bash
# tools/cin_pooled_fused_pmc.sh (GPU box): SQ counters of cin_pooled_k over tools/cin_pooled_fused_bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_cq -o cq -- python3 tools/cin_pooled_fused_bench.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_cq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = set()
for r in csv.DictReader(open(f)):
    if "cin_pooled_k" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
k = len(n)
m = {a: v / k for a, v in acc.items()}
cyc = m["GRBM_GUI_ACTIVE"] / 8
print(k, "launches; cycles %.0f  mfma busy %.3f  wait_any %.3f  wait_inst %.3f  active %.3f  valu insts %.3g  mfma %.3g" % (cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"], m["SQ_INSTS_MFMA"]))
PY
rm -rf gpurun_out/pmc_cq
