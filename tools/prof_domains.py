import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import safepy_amd
from safepy_amd import workloads
data = workloads.costanzo_surrogate(seed=0)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
sf.load_attributes(attribute_file=data['attributes'])
sf.compute_pvalues()
sf.define_top_attributes(); sf.define_domains()
for name, fn in (('define_top_attributes', sf.define_top_attributes), ('define_domains', sf.define_domains)):
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    print('=====', name)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
