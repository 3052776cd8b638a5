"""Diagnostic: time the phases of one bench step with a device sync after each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safepy_amd
from safepy_amd import backend as be, workloads

data = workloads.costanzo_surrogate(seed=0)
ctx = be.Context.default(0)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
t = time.perf_counter(); sf.define_neighborhoods(); ctx.sync(); print('define_neighborhoods %.2f ms' % (1e3 * (time.perf_counter() - t)))
nbr = sf._nbr
b = data['attributes']; n, m = b.shape
d_b = ctx.alloc(b.nbytes); d_b.upload(np.ascontiguousarray(b.T))
outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
P = 1000
table = be.nes_table(P)
for it in range(3):
    T = {}
    t0 = time.perf_counter()
    attr = be.Attributes.from_device(ctx, d_b.ptr, np.float32, n, m, order='F')
    st = attr.stats(); ctx.sync(); T['stats'] = time.perf_counter()
    flags = attr.row_flags(); T['flags'] = time.perf_counter()
    perms = be.Permutations(ctx, n, flags, P, 0); ctx.sync(); T['perms_create'] = time.perf_counter()
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs], table=table); ctx.sync(); T['randomization'] = time.perf_counter()
    perms.close(); attr.close(); T['close'] = time.perf_counter()
    prev = t0; line = []
    for k, v in T.items():
        line.append('%s %.2f' % (k, 1e3 * (v - prev))); prev = v
    print('iter', it, ' | '.join(line), '| total %.2f ms' % (1e3 * (prev - t0)), ctx.last_kernel())
# host-only: draw stream speed
import ctypes as C
vals = np.arange(3789, dtype=np.int64)
t = time.perf_counter(); be.rng_permutations_host(0, vals, 1000); print('host draw stream + host swaps for 1000 x 3789: %.2f ms' % (1e3 * (time.perf_counter() - t)))
