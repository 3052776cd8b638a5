"""The first compute_pvalues after define_neighborhoods, itemised (SAFE_HIP_TRACE=1 prints the library's own marks).
usage: SAFE_HIP_TRACE=1 python tools/probe/cold_call.py [cfg1|cfg3]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np                                   # noqa: E402
import safepy_amd                                    # noqa: E402
from safepy_amd import workloads                     # noqa: E402
import logging                                       # noqa: E402

logging.disable(logging.WARNING)
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
if which == 'cfg1':
    data = workloads.costanzo_surrogate(seed=0)
    graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
    metric, b, kw = 'shortpath_weighted_layout', data['attributes'], dict(how='randomization', num_permutations=1000)
else:
    n, m = 20000, 10000
    b = (np.random.default_rng(5).uniform(size=(n, m)) < 0.01).astype(np.float32)
    graph, metric, kw = safepy_amd.LayoutGraph(workloads.uniform_layout(4, n)), 'euclidean', {}
for rep in range(3):
    sf = safepy_amd.SAFE(verbose=False)
    sf.random_seed = 0
    sf.graph = graph
    t0 = time.perf_counter()
    sf.define_neighborhoods(node_distance_metric=metric, neighborhood_radius=0.1)
    t1 = time.perf_counter()
    sf.node2attribute = b
    print('--- rep %d: define_neighborhoods %.1f ms; first compute_pvalues:' % (rep, 1e3 * (t1 - t0)), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    sf.compute_pvalues(**kw)
    t1 = time.perf_counter()
    nes = np.asarray(sf.nes)
    t2 = time.perf_counter()
    sf.compute_pvalues(**kw)
    t3 = time.perf_counter()
    print('=== rep %d: first call %.1f ms, read nes %.1f ms, second call %.1f ms' % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), file=sys.stderr, flush=True)
    del sf, nes
