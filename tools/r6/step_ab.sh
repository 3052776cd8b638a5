#!/bin/bash
# round 6: driver-style seeded steps (5 + 20) under sets of environment switches, interleaved on one box
# usage (through gpurun): tools/r6/step_ab.sh <out tag> <repeats> "<name>=<ENV=val ENV=val ...>" ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; S=$O/summary.txt; : > $S; N=$2; shift 2
one() {  # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --extras 0 --cpu-perms 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['per_rank'][0]
print('%-28s mean %.3f  min/med/max %s  draw_busy %.2f  tables_enq %.2f  kbusy %.2f  unseeded %s' % ('$name', d['ms_per_step'], ' '.join('%.3f' % v for v in d['step_ms_min_median_max']),
      r['draw_busy_ms'], r['host_stream_ms'], r['gpu_kernel_busy_ms'], d.get('unseeded_device_stream', {}).get('1000_permutations', {}).get('ms_per_step')))" >> $S
}
for i in $(seq 1 $N); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    one "$name" $envs
  done
done
cat $S
