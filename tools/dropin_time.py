"""Wall-clock of the drop-in calls with NumPy in / NumPy out (the PCIe-inclusive figures of DESIGN.md
section 5): SAFE.define_neighborhoods() and SAFE.compute_pvalues() at configs[1] (permutation test)
and configs[3] (hypergeometric), with the phases of compute_pvalues timed separately.
usage: dropin_time.py [c2|c4]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safepy_amd
from safepy_amd import backend as be, workloads

which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
if which == 'c2':
    data = workloads.costanzo_surrogate(seed=0)
    graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
    b = data['attributes']
    kw = dict(how='randomization', num_permutations=1000)
    metric = 'shortpath_weighted_layout'
else:
    n, m = 20000, 10000
    graph = safepy_amd.LayoutGraph(workloads.uniform_layout(4, n))
    b = (np.random.default_rng(5).uniform(size=(n, m)) < 0.01).astype(np.float32)
    kw = {}
    metric = 'euclidean'
sf = safepy_amd.SAFE(verbose=False)
sf.random_seed = 0
sf.graph = graph
for it in range(3):
    t = time.perf_counter()
    sf.define_neighborhoods(node_distance_metric=metric, neighborhood_radius=0.1)
    t1 = time.perf_counter()
    sf.node2attribute = b
    sf.compute_pvalues(**kw)
    t2 = time.perf_counter()
    touched = float(sf.nes[0, 0]) + float(sf.nes_binary[0, 0]) + float(sf.pvalues_pos[0, 0])
    t3 = time.perf_counter()
    print('%s iter %d: define_neighborhoods %.1f ms | compute_pvalues %.1f ms | first touch of nes, nes_binary, pvalues_pos %.1f ms'
          % (which, it, 1e3 * (t1 - t), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
# phases
ctx = be.Context.default(0)
t = time.perf_counter(); attr = be.Attributes.from_host(ctx, b); ctx.sync(); t_up = time.perf_counter() - t
n, m = b.shape
buf = ctx.alloc_f64(n, m)
t = time.perf_counter(); h = buf.download((n, m)); t_dn = time.perf_counter() - t
t = time.perf_counter(); h2 = np.empty((n, m)); h2[:] = 1.0; t_alloc = time.perf_counter() - t
print('upload %s %.0f MB: %.1f ms (%.1f GB/s) | download f64 %.0f MB: %.1f ms (%.1f GB/s) | host alloc+touch of one output: %.1f ms'
      % (b.dtype, b.nbytes / 1e6, 1e3 * t_up, b.nbytes / t_up / 1e9, h.nbytes / 1e6, 1e3 * t_dn, h.nbytes / t_dn / 1e9, 1e3 * t_alloc))
