"""Kernels-only call (tables ready) under uniform launch spans: how many permutations should a launch / a task hold?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = sys.argv[1] if len(sys.argv) > 1 else '1000'
for span in ('', '128', '200', '250', '255', '334', '500'):
    cfg = {'SAFE_HIP_BITS_SPAN': span} if span else {}
    print('span', span or 'default', end=': ', flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bits_ablate.py'), '--one', P], env=dict(os.environ, SAFE_HIP_BITS_KERNEL='blk', **cfg))
