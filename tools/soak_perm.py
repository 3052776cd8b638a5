"""Soak of the permutation-test pipeline (draw thread, swap workers, table kernels, two kernel streams): repeats the seeded call
and compares every result matrix with the first call's, bit for bit.  usage: soak_perm.py [calls] [P]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safepy_amd
from safepy_amd import backend as be, workloads
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
data = workloads.costanzo_surrogate(seed=0)
ctx = be.Context.default(0)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
nbr = sf._nbr
b = data['attributes']; n, m = b.shape
attr = be.Attributes.from_host(ctx, b)
flags = attr.row_flags()
outs = [torch.empty((n, m), dtype=torch.float64, device='cuda') for _ in range(5)] + [torch.empty((m,), dtype=torch.float64, device='cuda')]
table = be.nes_table(P)
ref, bad, t0 = None, 0, time.perf_counter()
for it in range(calls):
    for o in outs:
        o.fill_(-7.0)
    torch.cuda.synchronize()
    perms = be.Permutations(ctx, n, flags, P, 0)
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.data_ptr() for o in outs], table=table)
    ctx.sync()
    perms.close()
    if ref is None:
        ref = [o.clone() for o in outs]
        assert not (ref[3] == -7.0).any()
    else:
        bad += sum(0 if torch.equal(a, r) else 1 for a, r in zip(outs, ref))
print('%d seeded calls (%d permutations) in %.1f s, mismatching matrices: %d' % (calls, P, time.perf_counter() - t0, bad))
sys.exit(1 if bad else 0)
