#!/bin/bash
# tools/r06_warm2.sh (GPU box): the lines of r06_warm.sh with bench.py's clock-settle phase (default 250 ms) and without it (DIR_BENCH_SETTLE_MS=0)
cd "$GRAFT_REPO_ROOT"
for wl in deepfm_gather_fm din mlp_dense cin; do
for ms in 250 0; do
  if [ $wl = cin ]; then a="--steps 5 --warmup 2"; elif [ $wl = deepfm_gather_fm ]; then a="--steps 20 --warmup 5"; else a="--steps 50 --warmup 5"; fi
  DIR_BENCH_SETTLE_MS=$ms DIR_BENCH_NO_SWEEP=1 timeout -k 10 300 python3 bench.py --workload $wl $a --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('secondary_cfg5_xdeepfm_cin') or {}
print('$wl $a settle $ms: ms_per_step %.4f  frac %.3f  median %.1f  settle %s  cfg5 %s %s' % (d['ms_per_step'], r['frac'], r.get('launch_us_median',0), (d['config'].get('clock_settle') or {}).get('untimed_steps_before_warmup'), s.get('ms_per_step'), s.get('per_gpu_frac_of_bf16_mfma_peak')))" || exit 1
done; done
