"""Binary randomization x 1000 permutations (unseeded: the kernels' own time) on the configs[1] surrogate's recipe scaled to N
nodes, M = 2048 attributes: call time, kernel, ps per member-word and permutation -- the cost scale of DESIGN section 7's
"off the fast path" figures.  python tools/r6/wide_n.py 3971 8100 8300 12000 20000 uniform20000
(SAFE_HIP_BITS_PRE=0 in the environment: without the pre-permuted forms -- what the shapes beyond N = 8190 ran before)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import safepy_amd                                      # noqa: E402
from safepy_amd import backend as be, workloads        # noqa: E402

ctx = safepy_amd.Context.default(0)
for arg in sys.argv[1:] or ['3971', '8100', '8300', '12000']:
    if arg.startswith('uniform'):                       # configs[3]'s network: uniform layout, euclidean r = 0.1 (577 members per node at 20 000)
        n = int(arg[7:])
        xy = workloads.uniform_layout(4, n)
        nbr = be.Neighborhoods.euclidean(ctx, xy, 0.1 * (xy[:, 0].max() - xy[:, 0].min()))
        b = np.asfortranarray((np.random.default_rng(5).uniform(size=(n, 2048)) < 0.01).astype(np.float32))
    else:
        n = int(arg)
        d = workloads.costanzo_surrogate(seed=1, n=n, m=2048, target_edges=int(28202 * n / 3971), n_nan_rows=int(182 * n / 3971))
        sf = safepy_amd.SAFE(verbose=False)
        sf.graph = safepy_amd.LayoutGraph(d['xy'], d['edge_u'], d['edge_v'], length=d['length'])
        sf.define_neighborhoods()
        nbr, b = sf._nbr, d['attributes']
    m = b.shape[1]
    attr = be.Attributes.from_host(ctx, b)
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    best = None
    for _ in range(3):
        perms = be.Permutations(ctx, n, attr.row_flags(), 1000, None)
        ctx.sync()
        t0 = time.perf_counter()
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs])
        ctx.sync()
        dt = time.perf_counter() - t0
        perms.close()
        name = ctx.last_kernel()[0]
        best = dt if best is None or dt < best else best
    print('%s N=%d: %s, call %.2f ms, members/node %.1f, %.3f ps per member-word and permutation'
          % (arg, n, name, 1e3 * best, nbr.nnz / float(n), 1e12 * best / (float(nbr.nnz) * ((m + 63) // 64) * 1000)), flush=True)
    for o in outs:
        o.free()
    attr.close()
