#!/bin/bash
# tools/r03_round_a.sh (GPU box): host CPU facts for the cpu_baseline leg, then the bench lines + kernel-trace summaries that round 2 left
# stale (dcn_full, esmm_full, multihot_bag) -> gpurun_out/
cd "$GRAFT_REPO_ROOT"
{ echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)";
  python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; lscpu | grep -E "^(Model name|Socket|Core|Thread|CPU\(s\)|NUMA)"; } > gpurun_out/r03_host_cpu.txt 2>&1
cat gpurun_out/r03_host_cpu.txt
b() { name=$1; shift; python3 bench.py "$@" > gpurun_out/bench_$name.log 2>&1 && grep '^{' gpurun_out/bench_$name.log | tail -1 > gpurun_out/r03_bench_$name.json; echo "$name: $(cut -c1-260 gpurun_out/r03_bench_$name.json)"; }
b multihot_bag --workload multihot_bag --steps 50 --warmup 5 --no-cpu-baseline
b dcn_full --workload dcn_full --steps 50 --warmup 5 --no-cpu-baseline
b esmm_full --workload esmm_full --steps 50 --warmup 5 --no-cpu-baseline
for w in multihot_bag dcn_full esmm_full; do
    bash tools/prof.sh $w -- --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1; head -8 gpurun_out/prof_$w.txt
done
