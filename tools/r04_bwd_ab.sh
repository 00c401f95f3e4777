#!/bin/bash
# tools/r04_bwd_ab.sh: same-box A/B of the fp16 x 2 backward kernels (DIR_DENSE_BWD_SPLIT / DIR_CIN_BWD_SPLIT = f16x2 | bf16x3) on the
# training-step workloads.  Output: gpurun_out/r04_bwd_ab.txt
mkdir -p gpurun_out
out=gpurun_out/r04_bwd_ab.txt
: > $out
for wl in deepfm_train dcn_train esmm_train xdeepfm_train cin_backward; do
  for sp in f16x2 bf16x3; do
    DIR_DENSE_BWD_SPLIT=$sp DIR_CIN_BWD_SPLIT=$sp python bench.py --workload $wl --steps 40 --warmup 8 > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "$wl $sp FAILED" >> $out; tail -3 gpurun_out/_ab.err >> $out; continue; }
    python - "$wl" "$sp" >> $out <<'PY'
import json,sys
d=json.load(open("gpurun_out/_ab.json"))
print("%-14s %-7s %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
cat $out
