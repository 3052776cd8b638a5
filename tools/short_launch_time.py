"""Kernel time of ONE launch of the bit-sliced permutation test at configs[1] for short permutation counts (the pipeline's first
and last stages): tables generated before the call.  usage: short_launch_time.py [P ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safepy_amd
from safepy_amd import backend as be, workloads
be.pin_threads_to_device_numa(0)
data = workloads.costanzo_surrogate(seed=0)
ctx = be.Context.default(0)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
nbr = sf._nbr
b = data['attributes']; n, m = b.shape
attr = be.Attributes.from_host(ctx, b)
attr.stats()
flags = attr.row_flags()
outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
for P in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 96, 128]:
    table = be.nes_table(P)
    res = []
    for it in range(6):
        perms = be.Permutations(ctx, n, flags, P, 0)
        perms.read(P - 1, P)
        ctx.sync(); t0 = time.perf_counter()
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs], table=table)
        ctx.sync(); dt = time.perf_counter() - t0
        name, ms, launches = ctx.last_kernel()
        perms.close()
        if it:
            res.append((ms * launches, launches, 1e3 * dt))
    k = min(r[0] for r in res)
    print('P=%4d: %d launch(es), kernel time %.3f ms = %.2f us per permutation; call %.3f ms' % (P, res[0][1], k, 1e3 * k / P, min(r[2] for r in res)), flush=True)
