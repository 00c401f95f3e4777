#!/bin/bash
# tools/r06_tower_cs.sh (GPU box): the tower tests on the column-split kernel, then mlp_dense / deepfm_full / esmm_full / xdeepfm_full with it and with tower_bf3_k, A B A B
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_tower.py tests/test_gpu_models.py tests/test_gpu_range.py -x -q > gpurun_out/_tower_cs_tests.txt 2>&1 || { tail -40 gpurun_out/_tower_cs_tests.txt; exit 1; }
tail -2 gpurun_out/_tower_cs_tests.txt
for wl in mlp_dense deepfm_full; do for k in cs rows cs rows; do
  DIR_TOWER_KERNEL=$k timeout -k 10 300 python3 bench.py --workload $wl --steps 200 --warmup 1000 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl kernel $k: ms_per_step %.4f  frac %.3f  median %.1f' % (d['ms_per_step'], r['frac'], r.get('launch_us_median',0)))" || exit 1
done; done
