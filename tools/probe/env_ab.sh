#!/bin/bash
# generic A/B of one environment variable over driver-style runs: env_ab.sh VAR "v1 v2 ..." [runs]
var=$1; vals=$2; n=${3:-12}
for i in $(seq $n); do
  for v in $vals; do
    env $var=$v timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); s = d['step_probe']['slowest_steps'][0]
        print('$var=$v mean %.3f median %.3f slowest: step %d %.2f ms faults %d' % (d['ms_per_step'], d['step_ms_min_median_max'][1], s['step'], s['ms'], s['minor_faults']))
"
  done
done | sort | awk -v var=$var '{print} {k=$1; n[k]++; m[k]+=$3; md[k]+=$5; if ($0 ~ /faults 4[0-9][0-9][0-9]/) q[k]++; if ($10+0 > 5.0) big[k]++} END{for (k in n) print "SUMMARY", k, n[k], "runs: mean of means", m[k]/n[k], "mean of medians", md[k]/n[k], "runs with a ~4100-fault step", q[k]+0, "runs with a step > 5 ms", big[k]+0}'
