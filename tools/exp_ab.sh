#!/bin/bash
# A/B of one environment switch on the headline step, interleaved on ONE box: exp_ab.sh VAR=a VAR=b [repeats]
A=$1; B=$2; N=${3:-3}
for rep in $(seq 1 $N); do
  for cfg in "$A" "$B"; do
    env $cfg python bench.py --steps 200 --extras 0 --cpu-perms 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['per_rank'][0]
print('%-34s mean %.3f median %.3f  stream %.2f draw %.2f kbusy %.2f' % ('$cfg', d['ms_per_step'], d['step_ms_min_median_max'][1], r['host_stream_ms'], r['draw_busy_ms'], r['gpu_kernel_busy_ms']))"
  done
done
