#!/bin/bash
# round 3, step e: DIN forward on bf16x3 -- parity tests, then the A/B bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x -k "din or DIN" > gpurun_out/r03_din_tests.log 2>&1; echo "din tests rc=$?"; tail -5 gpurun_out/r03_din_tests.log
for a in bf16x3 f32; do
  DIR_DIN_ARITH=$a timeout -k 10 300 python3 bench.py --workload din --steps 50 --warmup 10 > gpurun_out/r03_bench_din_$a.json 2> gpurun_out/bench_din_$a.log || echo "bench $a failed"
  echo "$a: $(cut -c1-400 gpurun_out/r03_bench_din_$a.json)"
done
