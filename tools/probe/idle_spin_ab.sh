#!/bin/bash
# Does the slow FIRST timed step of a 5 + 20 run come from the idle draw worker having gone to sleep during the pause between
# warm-up and timed region (its wake-up through a futex takes milliseconds now and then)?  A/B of the worker's idle polling window.
for i in $(seq ${1:-8}); do
  for us in 1500 20000; do
    SAFE_HIP_DRAW_IDLE_SPIN_US=$us timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-perms 0 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); s = d['step_probe']['slowest_steps'][0]; print('idle spin $us us: mean %.3f' % d['ms_per_step'], 'slowest: step', s['step'], '%.2f ms' % s['ms'], 'faults', s['minor_faults'], 'tables_enq %.2f' % s['tables_enqueued_ms'])
"
  done
done
