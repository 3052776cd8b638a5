import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import safepy_amd
from safepy_amd import workloads
data = workloads.costanzo_surrogate(seed=0)
sf = safepy_amd.SAFE(verbose=False)
sf.random_seed = 0
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.1)
sf.node2attribute = data['attributes']
for mt in (False, True, False, True):
    t = time.perf_counter(); sf.compute_pvalues(how='randomization', num_permutations=1000, multiple_testing=mt); dt = time.perf_counter() - t
    print('randomization multiple_testing=%s: %.2f ms' % (mt, 1e3 * dt))
for mt in (False, True, False, True):
    t = time.perf_counter(); sf.compute_pvalues(how='hypergeometric', multiple_testing=mt); dt = time.perf_counter() - t
    print('hypergeometric multiple_testing=%s: %.2f ms' % (mt, 1e3 * dt))
