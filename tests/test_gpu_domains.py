"""define_top_attributes / define_domains / trim_domains (the consumers of nes_binary,
safepy/safe.py:610-745) on the device path vs the real reference's outputs
(tests/golden/domains.npz) and vs the oracle on a larger seeded case.  Needs an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'domains.npz')


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


def _pipeline(amd, g, names=None):
    import pandas as pd
    sf = amd.SAFE(verbose=False)
    xy, eu, ev = g['xy'], g['edge_u'], g['edge_v']
    length = np.sqrt(((xy[eu] - xy[ev]) ** 2).sum(axis=1))
    sf.graph = amd.LayoutGraph(xy, eu, ev, length=length)
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    sf.load_attributes(attribute_file=g['attributes'].copy())
    if names is not None:
        sf.attributes = pd.DataFrame({'id': np.arange(len(names)), 'name': list(names)})
    sf.compute_pvalues()
    return sf


def test_reference_pipeline_golden(amd):
    g = dict(np.load(GOLDEN))
    sf = _pipeline(amd, g, names=g['names'])
    assert np.array_equal(sf.nes_binary, g['nes_binary'])
    sf.define_top_attributes()
    a = sf.attributes
    assert np.array_equal(a['top'].values, g['top'].astype(bool))
    assert np.array_equal(a['num_connected_components'].values, g['num_cc'])
    assert np.array_equal(a['num_large_connected_components'].values, g['num_large_cc'])
    for j, s in enumerate(a['size_connected_components']):
        want = g['cc_sizes'][j][g['cc_sizes'][j] >= 0]
        assert (s is None and len(want) == 0) or np.array_equal(s, want)
    for thr in (0.75, 0.65):
        tag = 'thr%g_' % thr
        sf.define_domains(attribute_distance_threshold=thr)
        assert np.array_equal(sf.attributes['domain'].values, g[tag + 'domain'])
        cols = [c for c in sf.node2domain.columns if c not in ('primary_domain', 'primary_nes')]
        assert np.array_equal(np.array(cols), g[tag + 'domain_ids'])
        assert np.array_equal(sf.node2domain[cols].values, g[tag + 'node2domain'])
        assert np.array_equal(sf.node2domain['primary_domain'].values, g[tag + 'primary_domain'])
        # NES of the hypergeometric path: same tolerance as the p-values they come from
        np.testing.assert_allclose(sf.node2domain['primary_nes'].values, g[tag + 'primary_nes'], rtol=1e-6, atol=1e-9)
    sf.trim_domains()
    assert np.array_equal(sf.attributes['domain'].values, g['trim_domain'])
    assert np.array_equal(sf.node2domain['primary_domain'].values, g['trim_primary_domain'])
    np.testing.assert_allclose(sf.node2domain['primary_nes'].values, g['trim_primary_nes'], rtol=1e-6, atol=1e-9)
    assert np.array_equal(sf.domains['id'].values, g['trim_domain_ids'])
    assert list(sf.domains['label'].values) == list(g['trim_domain_labels'])


def test_components_and_jaccard_vs_oracle_midsize(amd):
    """Many attributes at once: component labels (sizes, counts) vs networkx through the oracle,
    condensed Jaccard distances bit-identical to SciPy's pdist."""
    from scipy.spatial import cKDTree
    from scipy.spatial.distance import pdist
    from safepy_amd import backend as be
    rng = np.random.default_rng(91)
    n, m = 1500, 120
    xy = rng.uniform(size=(n, 2))
    pairs = cKDTree(xy).query_pairs(0.035, output_type='ndarray')
    eu, ev = pairs[:, 0], pairs[:, 1]
    nb = np.zeros((n, m))
    for j in range(m):
        c = xy[rng.integers(n)]
        nb[:, j] = (np.sqrt(((xy - c) ** 2).sum(1)) < rng.uniform(0.05, 0.25)) & (rng.uniform(size=n) < rng.uniform(0.3, 1.0))
    nb[:, 3] = 0                                                 # nothing enriched
    nb[:, 4] = 1                                                 # everything enriched
    ctx = amd.Context.default(0)
    labels = be.enriched_components(ctx, n, eu, ev, nb)
    want = orc.top_attributes(nb, np.full(m, n), n, eu, ev, min_size=10)
    for j in range(m):
        lab = labels[j]
        assert np.array_equal(lab >= 0, nb[:, j] > 0)
        sizes = np.sort(np.bincount(lab[lab >= 0]))[::-1]
        sizes = sizes[sizes > 0]
        ws = want['size_connected_components'][j]
        assert np.array_equal(sizes, ws if ws is not None else np.zeros(0, int)), j
        # every node carries the smallest id of its component
        roots = np.unique(lab[lab >= 0])
        assert all(lab[r] == r for r in roots)
    d = be.jaccard_condensed(ctx, nb.T)
    assert np.array_equal(d, pdist(nb.T, 'jaccard'))
