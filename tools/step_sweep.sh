#!/bin/bash
# Whole-step time of the headline bench under launch-plan knobs (run on the GPU box): SAFE_HIP_BITS_SPARE = CUs the
# persistent permutation kernel leaves to the table kernels of the next pipeline stage, SAFE_HIP_BITS_TASKS, SAFE_HIP_BITS_MERGE
for cfg in "SAFE_HIP_BITS_SPARE=0" "SAFE_HIP_BITS_SPARE=2" "SAFE_HIP_BITS_SPARE=4" "SAFE_HIP_BITS_SPARE=8" "SAFE_HIP_BITS_SPARE=12" "SAFE_HIP_BITS_SPARE=16" "SAFE_HIP_BITS_SPARE=32"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 30 --warmup 3 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['step_ms_min_median_max'], round(d['roofline']['kernel_ms'],4), d['kernel_share_of_step'])"
done
echo "== P=10000"; for cfg in "SAFE_HIP_BITS_SPARE=0" "SAFE_HIP_BITS_SPARE=4" "SAFE_HIP_BITS_SPARE=8" "SAFE_HIP_BITS_SPARE=16"; do env $cfg python bench.py --perms 10000 --steps 5 --warmup 2 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), d['step_ms_min_median_max'])"; done
