#!/bin/bash
# CUs left out of the bit-sliced kernel's grid for the table kernels of the next stage (SAFE_HIP_BITS_SPARE): step / kernels per value
for i in 1 2; do
for v in ${@:-8 12 16 24}; do
SAFE_HIP_BITS_SPARE=$v timeout 300 python3 bench.py --steps 100 --warmup 30 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['per_rank'][0]
        print('spare=$v: step mean %.3f median %.3f  kernels busy %.3f  draw busy %.3f' % (d['ms_per_step'], d['step_ms_min_median_max'][1], r['gpu_kernel_busy_ms'], r['draw_busy_ms']))
" < /dev/stdin
done; done
