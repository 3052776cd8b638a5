"""GPU-only time of the permutation test at configs[1] with the permutation tables generated BEFORE the call
(no host stream in the timed region): what the enrichment kernels alone need, per launch plan."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
P = 1000
table = be.nes_table(P)
for it in range(4):
    perms = be.Permutations(ctx, n, flags, P, 0)
    perms.read(P - 1, P)                      # the whole table is on the device now
    ctx.sync(); t0 = time.perf_counter()
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs], table=table)
    ctx.sync(); dt = time.perf_counter() - t0
    name, ms, launches = ctx.last_kernel()
    print('tables ready: call %.2f ms, %s %d launches x %.3f ms = %.2f ms' % (1e3 * dt, name, launches, ms, ms * launches))
    perms.close()
