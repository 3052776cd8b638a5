"""configs[4] rank share (20 000 x 6250 f64 x 1000 permutations) on the matrix-core kernel: filtered vs six-slice form.
usage: python tools/probe/mfma_share.py [m] [nperm] [score]      (env SAFE_HIP_MFMA_FILTER etc. apply)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np                                   # noqa: E402
from safepy_amd import backend as be, workloads     # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
nperm = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
score = sys.argv[3] if len(sys.argv) > 3 else 'sum'
n = 20000
ctx = be.Context.default(0)
xy = workloads.uniform_layout(4, n)
nbr = be.Neighborhoods.euclidean(ctx, xy, 0.1 * (xy[:, 0].max() - xy[:, 0].min()))
b = workloads.quantitative_attributes(3, n, m)
attr = be.Attributes.from_host(ctx, b)
del b
outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
keep = None
for variant in os.environ.get('VARIANTS', 'own,own,general,six').split(','):
    os.environ['SAFE_HIP_MFMA_FILTER'] = '0' if variant == 'six' else '1'
    os.environ['SAFE_HIP_MFMA_FORM'] = 'general' if variant == 'general' else 'own'
    perms = be.Permutations(ctx, n, attr.row_flags(), nperm, 0)
    ctx.sync()
    t0 = time.perf_counter()
    be.randomization(ctx, nbr, attr, perms, score, 'both', 0.05, [o.ptr for o in outs])
    ctx.sync()
    dt = time.perf_counter() - t0
    perms.close()
    name, ms, launches = ctx.last_kernel()
    core, und = be.last_mfma_filter(ctx)
    nes = outs[3].download((n, m))
    same = None if keep is None else bool(np.array_equal(nes, keep, equal_nan=True))
    keep = nes
    print('filter=%s %s: call %.1f ms, kernels %.1f ms in %d launches, core slices %d, undecided %d (%.2e of compares), nes == previous: %s'
          % (variant, name, 1e3 * dt, ms * launches, launches, core, und, und / (float(n) * m * nperm), same), flush=True)
