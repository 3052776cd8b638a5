#!/bin/bash
# head of the permutation pipeline's stage plan, A/B on one box (seeded headline step); usage: stage_sweep.sh <out file> <plan> ...
# (plan = comma-separated first boundaries, "default" = the built-in plan)
export GPU_MAX_HW_QUEUES=8
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${1:-stage_sweep.txt}; shift; : > $OUT; cd $R
run() { env "$@" timeout 300 python bench.py --steps ${STEPS:-60} --warmup 4 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), [round(x,3) for x in d['step_ms_min_median_max']], round(d['roofline']['kernel_ms']*d['roofline']['launches_per_step'],3))" >> $OUT; }
for rep in 1 2 3 4; do
  for plan in "$@"; do
    if [ "$plan" = default ]; then run SAFE_X=default; else run SAFE_HIP_STAGES=$plan; fi
  done
done
sort $OUT | awk '{k=$1; s[k]+=$2; n[k]++; m[k]=m[k]" "$2} END {for (k in s) printf "%-40s mean %.3f  (%s )\n", k, s[k]/n[k], m[k]}'
