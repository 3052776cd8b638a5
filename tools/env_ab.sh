#!/bin/bash
# A/B of environment switches on the seeded headline step, interleaved repetitions on one box.
# usage: env_ab.sh <out file> "<ENV=VAL[ ENV=VAL]>" ...      ("X=0" = the defaults);  STEPS, REPS, PERMS from the environment
export GPU_MAX_HW_QUEUES=8
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${1:-env_ab.txt}; shift; : > $OUT; cd $R
run() { env $1 timeout 300 python bench.py --perms ${PERMS:-1000} --steps ${STEPS:-60} --warmup 4 --cpu-perms 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.replace(' ','+'), round(d['ms_per_step'],3), [round(x,3) for x in d['step_ms_min_median_max']], round(d['roofline']['kernel_ms']*d['roofline']['launches_per_step'],3))" >> $OUT; }
for rep in $(seq 1 ${REPS:-4}); do
  for v in "$@"; do run "$v"; done
done
sort $OUT | awk '{k=$1; s[k]+=$2; n[k]++; m[k]=m[k]" "$2; ks[k]+=$NF} END {for (k in s) printf "%-50s mean step %.3f  kernels %.3f  (%s )\n", k, s[k]/n[k], ks[k]/n[k], m[k]}'
