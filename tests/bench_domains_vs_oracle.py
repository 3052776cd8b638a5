"""define_top_attributes / define_domains on the config-2 surrogate (3971 nodes x 4373 GO-like binary
attributes, hypergeometric p-values): device path vs the oracle's networkx / SciPy calls (what the
reference does) on the same host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safepy_amd
from safepy_amd import workloads, backend as be
from oracle import safe_oracle as orc

data = workloads.costanzo_surrogate(seed=0)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
sf.load_attributes(attribute_file=data['attributes'])
t = time.perf_counter(); sf.compute_pvalues(); print('compute_pvalues (hypergeometric, incl. PCIe): %.1f ms' % (1e3 * (time.perf_counter() - t)))
n, m = sf.nes_binary.shape
for it in range(2):
    t = time.perf_counter(); sf.define_top_attributes(); t1 = time.perf_counter() - t
    n_cand = int((sf.attributes['num_neighborhoods_enriched'] >= 10).sum())
    n_top = int(sf.attributes['top'].sum())
    t = time.perf_counter(); sf.define_domains(); t2 = time.perf_counter() - t
    print('device: define_top_attributes %.1f ms (%d candidates -> %d top), define_domains %.1f ms (%d domains)'
          % (1e3 * t1, n_cand, n_top, 1e3 * t2, sf.attributes['domain'].max()))
ctx = be.Context.default(0)
top = sf.attributes['top'].values
x = np.ascontiguousarray(sf.nes_binary[:, top].T)
t = time.perf_counter(); d = be.jaccard_condensed(ctx, x); tj = time.perf_counter() - t
from scipy.spatial.distance import pdist
t = time.perf_counter(); dref = pdist(x, 'jaccard'); tjr = time.perf_counter() - t
print('jaccard condensed (%d profiles x %d nodes): device %.1f ms (incl. upload/download), scipy pdist %.1f ms, identical: %s'
      % (x.shape[0], x.shape[1], 1e3 * tj, 1e3 * tjr, np.array_equal(d, dref)))
t = time.perf_counter()
want = orc.top_attributes(sf.nes_binary, sf.attributes['num_neighborhoods_enriched'].values, n, data['edge_u'], data['edge_v'], 10)
tr = time.perf_counter() - t
print('oracle (networkx subgraph + connected_components per attribute): %.1f ms; same top set: %s'
      % (1e3 * tr, np.array_equal(want['top'], top)))
