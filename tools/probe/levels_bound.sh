#!/bin/bash
# upper bound of what fewer counter levels could gain in k_permtest_bits_blk (diagnostic builds, wrong results): make DIAG=1 first
cd $GRAFT_REPO_ROOT
for d in 0 2048 1024 0; do
  echo "SAFE_HIP_BITS_DBG=$d"
  SAFE_HIP_BITS_DBG=$d python3 bench.py --steps 60 --warmup 20 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  step mean %.3f median %.3f  kernel_ms/launch %.4f  busy/step %.3f' % (d['ms_per_step'], d['step_ms_min_median_max'][1], r.get('kernel_ms',0), r.get('kernel_busy_ms_per_step',0)))
"
done
