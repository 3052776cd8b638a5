"""Kernels-only call at configs[1] with P permutations ready on the device, for several launch plans (SAFE_HIP_BITS_MERGE,
SAFE_HIP_BITS_TASKS): how long should a launch be when the tables are not the bottleneck (unseeded runs, many permutations)?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = sys.argv[1] if len(sys.argv) > 1 else '10000'
for cfg in ({}, {'SAFE_HIP_BITS_MERGE': '2'}, {'SAFE_HIP_BITS_MERGE': '3'}, {'SAFE_HIP_BITS_MERGE': '4'}, {'SAFE_HIP_BITS_MERGE': '2', 'SAFE_HIP_BITS_TASKS': '3'},
            {'SAFE_HIP_BITS_MERGE': '2', 'SAFE_HIP_BITS_TASKS': '4'}, {'SAFE_HIP_BITS_MERGE': '4', 'SAFE_HIP_BITS_TASKS': '4'}):
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bits_ablate.py'), '--one', P], env=dict(os.environ, SAFE_HIP_BITS_KERNEL='blk', **cfg))
