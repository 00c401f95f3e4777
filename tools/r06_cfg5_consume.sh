#!/bin/bash
# tools/r06_cfg5_consume.sh (GPU box): the gather-form tests, then the config-5 leg on one rank with its collectives issued (DIR_BENCH_CFG5_SHARDED=1),
# rows form (no finish pass) against lookup_async(out=) + the plain layers, A B A B
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_cin_gather.py -x -q > gpurun_out/_cin_gather_tests.txt 2>&1 || { tail -20 gpurun_out/_cin_gather_tests.txt; exit 1; }; tail -2 gpurun_out/_cin_gather_tests.txt
for c in 1 0 1 0; do
  DIR_BENCH_CFG5_SHARDED=1 DIR_BENCH_CFG5_CONSUME=$c DIR_BENCH_NO_SWEEP=1 timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary_cfg5_xdeepfm_cin']
print('consume $c: cfg5 ms_per_step %.4f  cin_only %.4f  frac %.3f  parity %s' % (s['ms_per_step'], s['cin_only_ms_per_step'], s['per_gpu_frac_of_bf16_mfma_peak'], (s.get('parity_check') or {}).get('ok')))
print('   ', s['lookup'][:200])" || exit 1
done
