run() { env "$@" python bench.py --steps 30 --warmup 3 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), [round(x,3) for x in d['step_ms_min_median_max']], round(d['roofline']['kernel_ms']*d['roofline']['launches_per_step'],3))"; }
for t in 1 2 3; do run SAFE_HIP_BITS_TASKS=$t; done
for sp in 48 64 96 128 160; do for t in 1 2; do run SAFE_HIP_BITS_SPAN=$sp SAFE_HIP_BITS_TASKS=$t; done; done
run10() { env "$@" python bench.py --perms 10000 --steps 5 --warmup 2 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), [round(x,3) for x in d['step_ms_min_median_max']])"; }
for sp in 64 96 128; do for t in 1 2; do run10 SAFE_HIP_BITS_SPAN=$sp SAFE_HIP_BITS_TASKS=$t; done; done
run10 SAFE_HIP_BITS_TASKS=1
