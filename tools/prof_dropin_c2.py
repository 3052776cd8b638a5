import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import safepy_amd
from safepy_amd import workloads
which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
if which == 'c2':
    data = workloads.costanzo_surrogate(seed=0)
    graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
    b = data['attributes']; kw = dict(how='randomization', num_permutations=1000); metric = 'shortpath_weighted_layout'
else:
    n, m = 20000, 10000
    graph = safepy_amd.LayoutGraph(workloads.uniform_layout(4, n))
    b = (np.random.default_rng(5).uniform(size=(n, m)) < 0.01).astype(np.float32); kw = {}; metric = 'euclidean'
sf = safepy_amd.SAFE(verbose=False)
sf.random_seed = 0
sf.graph = graph
sf.define_neighborhoods(node_distance_metric=metric, neighborhood_radius=0.1)
sf.node2attribute = b
import logging; logging.disable(logging.WARNING)
for _ in range(3): sf.compute_pvalues(**kw)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): sf.compute_pvalues(**kw)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
