"""Where the time of the Example-3 shape goes (1586 nodes x 1 quantitative attribute x 10000 permutations)."""
import os, sys, time, tempfile, logging
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import safepy_amd
from safepy_amd import workloads, backend as be
logging.disable(logging.WARNING)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, 'x.scatter')
keys, xy, att = workloads.example3_scatter(path)
for seed in (0, None):
    sf = safepy_amd.SAFE(verbose=False); sf.random_seed = seed
    sf.load_network(network_file=path, node_key_attribute='key')
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.06)
    sf.load_attributes(attribute_file=att)
    for rep in range(4):
        t0 = time.perf_counter(); sf.compute_pvalues(num_permutations=10000); t1 = time.perf_counter()
        ctx = be.Context.default(0)
        name, ms, launches = ctx.last_kernel()
        print('seed', seed, 'call %.2f ms; dominant kernel %s: %.3f ms x %d launches, busy %.2f ms' % (1e3 * (t1 - t0), name, ms, launches, ctx.last_kernel_busy_ms()))
