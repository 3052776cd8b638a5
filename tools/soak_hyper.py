"""Soak of the split hypergeometric form (two streams, three kernels): repeats the call and compares every result matrix with
the first call's, bit for bit.  usage: soak_hyper.py [calls] [M]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safepy_amd import backend as be, workloads
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = int(sys.argv[2]) if len(sys.argv) > 2 else 3001
n = 20000
ctx = be.Context.default(0)
xy = workloads.uniform_layout(4, n)
nbr = be.Neighborhoods.euclidean(ctx, xy, 0.1 * (xy[:, 0].max() - xy[:, 0].min()))
rng = np.random.default_rng(5)
b = (rng.uniform(size=(n, m)) < 0.01).astype(np.float32)
b[rng.choice(n, 500, replace=False)] = np.nan
attr = be.Attributes.from_host(ctx, b)
outs = [torch.empty((n, m), dtype=torch.float64, device='cuda') for _ in range(3)] + [torch.empty((m,), dtype=torch.float64, device='cuda')]
ref = None
t0 = time.perf_counter()
bad = 0
for it in range(calls):
    for o in outs:
        o.fill_(-7.0)                                   # stale values must not survive
    torch.cuda.synchronize()
    be.hypergeom(ctx, nbr, attr, 0.05, [o.data_ptr() for o in outs])
    ctx.sync()
    if ref is None:
        ref = [o.clone() for o in outs]
        assert ctx.last_kernel()[0] == 'k_hyp_emit'
        assert not (ref[0] == -7.0).any()
    else:
        for a, r in zip(outs, ref):
            if not torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(r, nan=-1.0)):
                bad += 1
print('%d calls of %d x %d in %.1f s, mismatching matrices: %d' % (calls, n, m, time.perf_counter() - t0, bad))
sys.exit(1 if bad else 0)
