"""Quantitative-attribute variant of the config-2 shape (doxorubicin-like real values, f64):
times the general f64 permutation kernels (LDS-resident vs global-tile gather) and z-score."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safepy_amd
from safepy_amd import backend as be, workloads

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 200
data = workloads.costanzo_surrogate(seed=0, m=8)
ctx = be.Context.default(0)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
nbr = sf._nbr
n = nbr.n
b = workloads.quantitative_attributes(3, n, m)
outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
for score in ('sum', 'z-score'):
    for path in ('lds', 'gather'):
        if path == 'gather':
            os.environ['SAFE_HIP_FORCE_PATH'] = 'gather'
        else:
            os.environ.pop('SAFE_HIP_FORCE_PATH', None)
        for it in range(2):
            attr = be.Attributes.from_host(ctx, b)
            perms = be.Permutations(ctx, n, attr.row_flags(), P, 0)
            ctx.sync(); t = time.perf_counter()
            be.randomization(ctx, nbr, attr, perms, score, 'both', 0.05, [o.ptr for o in outs]); ctx.sync()
            dt = time.perf_counter() - t
            perms.close(); attr.close()
        name, kms, kl = ctx.last_kernel()
        print('%-8s %-7s %s: call %.1f ms, kernel %.2f ms x %d -> %.3g enrichments/s (n=%d m=%d P=%d)'
              % (score, path, name, 1e3 * dt, kms, kl, n * m * P / dt, n, m, P))
