"""cProfile of the drop-in SAFE.compute_pvalues() / define_neighborhoods() at configs[1] (host-side overheads)."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import safepy_amd
from safepy_amd import workloads
data = workloads.costanzo_surrogate(seed=0)
graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
b = data['attributes']
sf = safepy_amd.SAFE(verbose=False)
sf.random_seed = 0
sf.graph = graph
for it in range(2):
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.1)
    sf.node2attribute = b
    sf.compute_pvalues(how='randomization', num_permutations=1000)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.1)
t1 = time.perf_counter()
sf.compute_pvalues(how='randomization', num_permutations=1000)
t2 = time.perf_counter()
pr.disable()
print('define_neighborhoods %.2f ms, compute_pvalues %.2f ms' % (1e3 * (t1 - t), 1e3 * (t2 - t1)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
