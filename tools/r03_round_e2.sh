timeout -k 10 300 python3 tools/din_bf3_probe.py > gpurun_out/r03_din_probe.txt 2>&1; cat gpurun_out/r03_din_probe.txt | tail -20
for a in bf16x3 f32; do
  DIR_DIN_ARITH=$a timeout -k 10 300 python3 bench.py --workload din --steps 50 --warmup 10 > gpurun_out/r03_bench_din_$a.json 2> gpurun_out/bench_din_$a.log || echo "bench $a failed"
  echo "$a: $(cut -c1-300 gpurun_out/r03_bench_din_$a.json)"
done
