#!/bin/bash
# A/B of the seeded step's host jitter.  usage (through gpurun): bash tools/probe/jitter_ab.sh <rounds> [steps] [warmup]
cd $GRAFT_REPO_ROOT
R=${1:-3}; S=${2:-200}; W=${3:-50}
show() { python3 - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    mn, med, mx = d.get('step_ms_min_median_max', [0, 0, 0])
    print('  mean %.3f  min/median/max %.3f/%.3f/%.3f  mean/median %.3f  host_cpu %.2f  slowest %s' % (
        d['ms_per_step'], mn, med, mx, d['ms_per_step'] / med if med else 0, d.get('host_cpu_ms_per_step', 0),
        [round(x, 2) for x in d.get('step_ms_slowest3', [])]), d.get('draw_threads', {}).get('chunks_won_by_twin_per_step_mean'))
PY
}
for r in $(seq 1 $R); do
  for v in ${VARIANTS:-old worker worker+cores}; do
    unset SAFE_BENCH_DRAW_CORES SAFE_HIP_DRAW_THREAD SAFE_HIP_DRAW_IDLE_SPIN_US SAFE_HIP_DRAW_TWIN
    case $v in
      twin) export SAFE_HIP_DRAW_TWIN=1;;
      notwin) ;;
      old) export SAFE_BENCH_DRAW_CORES=0 SAFE_HIP_DRAW_THREAD=percall SAFE_HIP_DRAW_TWIN=0;;
      worker) export SAFE_BENCH_DRAW_CORES=0;;
      worker+cores) ;;
      worker+cores1) export SAFE_BENCH_DRAW_CORES=1;;
      worker+cores4) export SAFE_BENCH_DRAW_CORES=4;;
      worker-nospin) export SAFE_BENCH_DRAW_CORES=0 SAFE_HIP_DRAW_IDLE_SPIN_US=0;;
    esac
    echo "round $r $v ($W + $S)"
    python3 bench.py --steps $S --warmup $W --extras 0 --cpu-perms 0 > /tmp/j.log 2>/tmp/j.err || tail -5 /tmp/j.err; show /tmp/j.log
  done
done
