#!/bin/bash
# tools/r04_bwd_ab2.sh: eager and --graph A/B of DIR_DENSE_BWD_SPLIT on the DeepFM / DCN / ESMM training steps -> gpurun_out/r04_bwd_ab2.txt
mkdir -p gpurun_out
out=gpurun_out/r04_bwd_ab2.txt
: > $out
for wl in deepfm_train esmm_train dcn_train; do
  for sp in f16x2 bf16x3; do
    for gr in "" "--graph"; do
      DIR_DENSE_BWD_SPLIT=$sp python bench.py --workload $wl --steps 40 --warmup 8 $gr > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "$wl $sp $gr FAILED" >> $out; tail -3 gpurun_out/_ab.err >> $out; continue; }
      python - "$wl" "$sp" "$gr" >> $out <<'PY'
import json,sys
d=json.load(open("gpurun_out/_ab.json"))
print("%-14s %-7s %-8s %.4f ms" % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"]))
PY
    done
  done
done
cat $out
