"""Kernels-only call at configs[1] (tools/bits_ablate.py --one) under the launch-plan knobs of the bit-sliced kernel, one child process each."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = sys.argv[1] if len(sys.argv) > 1 else '1000'
cfgs = [{}, {'SAFE_HIP_BITS_MINPPT': '8'}, {'SAFE_HIP_BITS_MINPPT': '32'}, {'SAFE_HIP_BITS_TARGETMIN': '128'}, {'SAFE_HIP_BITS_TARGETMIN': '512'},
        {'SAFE_HIP_BITS_SPARE': '8'}, {'SAFE_HIP_BITS_SPARE': '24'}, {'SAFE_HIP_BITS_SPARE': '0'}, {'SAFE_HIP_BITS_XCDQ': '0'}, {'SAFE_HIP_BITS_DBG': '2'},
        {'SAFE_HIP_BITS_OCC': '5'}, {}]
for cfg in cfgs:
    print(cfg, end=' ', flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bits_ablate.py'), '--one', P], env=dict(os.environ, SAFE_HIP_BITS_KERNEL='blk', **cfg))
