#!/bin/bash
# Whole-step time of the headline bench under the launch-plan knobs, after the draw stream stopped being the bottleneck:
# SAFE_HIP_BITS_MERGE (stages per launch after the start-up), SAFE_HIP_BITS_TASKS (queue depth per workgroup slot),
# SAFE_HIP_BITS_SPARE (CUs left to the table kernels).  Run on the GPU box.
run() { env "$@" python bench.py --steps 30 --warmup 3 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), [round(x,3) for x in d['step_ms_min_median_max']], round(d['roofline']['kernel_ms']*d['roofline']['launches_per_step'],3))"; }
for m in 1 2 3 4; do for t in 2 4 6; do run SAFE_HIP_BITS_MERGE=$m SAFE_HIP_BITS_TASKS=$t; done; done
for s in 0 4 16; do run SAFE_HIP_BITS_MERGE=2 SAFE_HIP_BITS_SPARE=$s; done
echo "== P=10000"
run10() { env "$@" python bench.py --perms 10000 --steps 5 --warmup 2 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), [round(x,3) for x in d['step_ms_min_median_max']])"; }
for m in 1 2 4; do for t in 2 6; do run10 SAFE_HIP_BITS_MERGE=$m SAFE_HIP_BITS_TASKS=$t; done; done
