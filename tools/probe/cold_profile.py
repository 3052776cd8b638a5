"""Python-level profile of the process's first SAFE.compute_pvalues (where the time outside the library's own marks goes)."""
import cProfile
import os
import pstats
import sys
import time
import logging

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np                                   # noqa: E402
import safepy_amd                                    # noqa: E402
from safepy_amd import workloads                     # noqa: E402

logging.disable(logging.WARNING)
data = workloads.costanzo_surrogate(seed=0)
graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf = safepy_amd.SAFE(verbose=False)
sf.random_seed = 0
sf.graph = graph
t0 = time.perf_counter()
sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.1)
print('define_neighborhoods (process first) %.1f ms' % (1e3 * (time.perf_counter() - t0)))
sf.node2attribute = data['attributes']
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
sf.compute_pvalues(how='randomization', num_permutations=1000)
pr.disable()
print('first compute_pvalues %.1f ms' % (1e3 * (time.perf_counter() - t0)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
