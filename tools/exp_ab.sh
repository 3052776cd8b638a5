#!/bin/bash
# A/B of environment switches on the headline step, interleaved on ONE box: exp_ab.sh "VAR=a" "VAR=b VAR2=c" ... (REPEATS=n, SEED=none for the unseeded call)
N=${REPEATS:-3}
for rep in $(seq 1 $N); do
  for cfg in "$@"; do
    env $cfg python - <<PY
import json, os, subprocess, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '200', '--extras', '1' if os.environ.get('SEED') == 'none' else '0', '--cpu-perms', '0'],
                     capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1]); r = d['per_rank'][0]
u = d.get('unseeded_device_stream', {})
print('%-44s seeded mean %.3f median %.3f  stream %.2f kbusy %.2f  unseeded %s / %s' % ("$cfg", d['ms_per_step'], d['step_ms_min_median_max'][1], r['host_stream_ms'], r['gpu_kernel_busy_ms'],
      '%.3f' % u['1000_permutations']['ms_per_step'] if u else '-', '%.2f' % u['10000_permutations']['ms_per_step'] if u else '-'))
PY
  done
done
