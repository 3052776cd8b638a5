import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import safepy_amd
from safepy_amd import workloads
n, m = 20000, 10000
graph = safepy_amd.LayoutGraph(workloads.uniform_layout(4, n))
b = (np.random.default_rng(5).uniform(size=(n, m)) < 0.01).astype(np.float32)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = graph
sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
sf.node2attribute = b
sf.compute_pvalues()
pr = cProfile.Profile(); pr.enable()
sf.compute_pvalues()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
