#!/usr/bin/env python3
"""N consecutive headline runs of bench.py: per-run mean / min / median / max step and what the per-step probe saw in the
slowest step (bench.py StepProbe).  usage: exp_outliers.py [N] [extra bench args ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
extra = sys.argv[2:]
for rep in range(n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '40', '--extras', '0', '--cpu-perms', '0'] + extra,
                         capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    mn, md, mx = d['step_ms_min_median_max']
    slow = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['step_probe']['slowest_steps'][0].items() if v}
    print('run %d mean %.3f min %.3f median %.3f max %.3f  max/median %.2f mean/median %.3f cpu/step %.1f' % (
        rep, d['ms_per_step'], mn, md, mx, mx / md, d['ms_per_step'] / md, d['host_cpu_ms_per_step']))
    print('    slowest:', slow, flush=True)
