"""Quantitative permutation test (matrix-core / f64 kernels) on the configs[1] surrogate's network scaled to N nodes: M normal f64
attributes x 1000 unseeded permutations, 'sum' and 'z-score': call time, kernel, fs per member x attribute x permutation.
python tools/r6/quant_n.py 3971:4373 3971:256 8300:2048 20000:2048"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import safepy_amd                                      # noqa: E402
from safepy_amd import backend as be, workloads        # noqa: E402

ctx = safepy_amd.Context.default(0)
for arg in sys.argv[1:] or ['3971:4373', '3971:256']:
    n, m = (int(v) for v in arg.split(':'))
    d = workloads.costanzo_surrogate(seed=1, n=n, m=64, target_edges=int(28202 * n / 3971), n_nan_rows=int(182 * n / 3971))
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(d['xy'], d['edge_u'], d['edge_v'], length=d['length'])
    sf.define_neighborhoods()
    nbr = sf._nbr
    b = np.asfortranarray(np.random.default_rng(3).normal(size=(n, m)))
    attr = be.Attributes.from_host(ctx, b)
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    for score in ('sum', 'z-score'):
        best = None
        for _ in range(3):
            perms = be.Permutations(ctx, n, attr.row_flags(), 1000, None)
            ctx.sync()
            t0 = time.perf_counter()
            be.randomization(ctx, nbr, attr, perms, score, 'both', 0.05, [o.ptr for o in outs])
            ctx.sync()
            dt = time.perf_counter() - t0
            perms.close()
            name = ctx.last_kernel()[0]
            best = dt if best is None or dt < best else best
        print('N=%d M=%d %s: %s, call %.2f ms, members/node %.1f, %.1f fs per member x attribute x permutation, filter %s'
              % (n, m, score, name, 1e3 * best, nbr.nnz / float(n), 1e15 * best / (float(nbr.nnz) * m * 1000), be.last_mfma_filter(ctx)), flush=True)
    for o in outs:
        o.free()
    attr.close()
