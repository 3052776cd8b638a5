#!/bin/bash
# A/B of two builds of the library on one box (both .so files under safepy_amd/; the second stays installed): ab_lib.sh <a.so> <b.so> [repeats]
A=$1; B=$2; N=${3:-3}
one() { python bench.py --extras 0 --cpu-perms 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['per_rank'][0]; print('$1 mean %.3f median %.3f kbusy %.2f' % (d['ms_per_step'], d['step_ms_min_median_max'][1], r['gpu_kernel_busy_ms']))"; }
for i in $(seq 1 $N); do
  cp safepy_amd/$A safepy_amd/libsafe_hip.so; one $A
  cp safepy_amd/$B safepy_amd/libsafe_hip.so; one $B
done
