#!/bin/bash
# tools/r06_din_ab.sh (GPU box): the cfg-4 DIN unit, packed kernel (round 6) against the wave-per-sample kernel, same box; kernel traces of both.
export ROUND=r06
python3 bench.py --workload din --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r06_bench_din_packed.json 2> gpurun_out/r06_bench_din_packed.err
DIR_DIN_PACKED=0 python3 bench.py --workload din --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r06_bench_din_wave.json 2> gpurun_out/r06_bench_din_wave.err
for w0 in 0 3 8 12; do
  DIR_DIN_PACK_W0=$w0 python3 bench.py --workload din --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r06_bench_din_packed_w$w0.json 2>/dev/null
done
bash tools/prof.sh din_packed -- --workload din --steps 50 --warmup 5 --no-cpu-baseline
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_bench_din_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print("%-50s ms_per_step %.4f  median launch %.1f us  p10 %.1f" % (f, d["ms_per_step"], r.get("launch_us_median", 0), r.get("launch_us_p10", 0)))
    except Exception as e:
        print(f, "unreadable", e)
PY
