#!/bin/bash
# A/B of the blocked bit-sliced kernel's per-XCD task queues (SAFE_HIP_BITS_XCDQ=0: tasks dealt to the queues one by one)
for x in 0 1 0 1; do
  export SAFE_HIP_BITS_XCDQ=$x
  python tools/bits_ablate.py --one 1000 2>/dev/null
  python bench.py --steps 60 --warmup 5 --cpu-perms 0 --extras 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  xcdq=$x P=1000 step', round(d['ms_per_step'],3), [round(v,2) for v in d['step_ms_min_median_max']])"
  python bench.py --steps 20 --warmup 3 --cpu-perms 0 --extras 0 --perms 10000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  xcdq=$x P=10000 step', round(d['ms_per_step'],3), [round(v,2) for v in d['step_ms_min_median_max']])"
done
