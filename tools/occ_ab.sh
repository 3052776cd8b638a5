#!/bin/bash
# A/B of the five-waves-per-SIMD build of the blocked bit-sliced kernel (SAFE_HIP_BITS_OCC=5) against the four-wave one:
# kernels only (tables ready), whole bench steps at 1000 and 10 000 permutations.
for occ in 0 5 0 5; do
  if [ $occ = 5 ]; then export SAFE_HIP_BITS_OCC=5; else export SAFE_HIP_BITS_OCC=4; fi
  python tools/bits_ablate.py --one 1000
  python bench.py --steps 60 --warmup 5 --cpu-perms 0 --extras 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  occ=$occ P=1000 step', round(d['ms_per_step'],3), d['step_ms_min_median_max'])"
  python bench.py --steps 20 --warmup 3 --cpu-perms 0 --extras 0 --perms 10000 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  occ=$occ P=10000 step', round(d['ms_per_step'],3), d['step_ms_min_median_max'])"
done
