#!/bin/bash
# tools/r04_carry_ab.sh: same-box A/B of DIR_DENSE_BWD_CARRY (the backward kernels' scales from the producing kernel's epilogue vs a max pass
# of their own), graph replays and eager -> gpurun_out/r04_carry_ab.txt
mkdir -p gpurun_out; out=gpurun_out/r04_carry_ab.txt; : > $out
for wl in deepfm_train esmm_train; do for c in 1 0; do for gr in "--graph" ""; do
  DIR_DENSE_BWD_CARRY=$c python bench.py --workload $wl --steps 60 --warmup 8 --no-cpu-baseline $gr 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-13s carry=%s %-8s %.4f ms' % ('$wl', '$c', '$gr', d['ms_per_step']))" >> $out
done; done; done
cat $out
