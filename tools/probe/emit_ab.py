"""k_hyp_emit at configs[3] (20 000 x 10 000 binary): 16-byte pair stores vs 8-byte stores (SAFE_HIP_EMIT_PAIR=0)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np                                   # noqa: E402
from safepy_amd import backend as be, workloads     # noqa: E402

n, m = 20000, 10000
ctx = be.Context.default(0)
xy = workloads.uniform_layout(4, n)
nbr = be.Neighborhoods.euclidean(ctx, xy, 0.1 * (xy[:, 0].max() - xy[:, 0].min()))
b = (np.random.default_rng(5).uniform(size=(n, m)) < 0.01).astype(np.float32)
attr = be.Attributes.from_host(ctx, b)
outs = [ctx.alloc_f64(n, m) for _ in range(3)] + [ctx.alloc_f64(m)]
keep = None
for pair in ('1', '0', '1', '0'):
    os.environ['SAFE_HIP_EMIT_PAIR'] = pair
    ts, ks = [], []
    for _ in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        be.hypergeom(ctx, nbr, attr, 0.05, [o.ptr for o in outs])
        ctx.sync()
        ts.append(1e3 * (time.perf_counter() - t0))
        name, ms, launches = ctx.last_kernel()
        ks.append(ms)
    p = outs[0].download((n, m))
    same = None if keep is None else bool(np.array_equal(p, keep))
    keep = p
    print('pair=%s: %s %.4f ms (median of 6; %.2f TB/s of 5.21 GB), call %.3f ms, p == previous: %s'
          % (pair, name, np.median(ks), 5.21e9 / (np.median(ks) * 1e-3) / 1e12, np.median(ts), same), flush=True)
