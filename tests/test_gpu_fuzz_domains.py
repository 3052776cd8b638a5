"""Random cases of the two device kernels behind define_top_attributes / define_domains (safe.py:610-700): connected components
of the enriched nodes of every attribute (sizes against networkx through the oracle, every node labelled with the smallest id of
its component) and condensed Jaccard distances between the attributes' enrichment columns (bit-identical to SciPy's pdist,
NaN for two empty columns included).

SAFE_FUZZ_SECONDS (default 15) bounds the run; SAFE_FUZZ_FIRST names the first case."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


def test_random_components_and_jaccard_against_the_oracle():
    from scipy.spatial.distance import pdist
    import safepy_amd as amd
    from safepy_amd import backend as be
    assert amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    ctx = amd.Context.default(0)
    budget = float(os.environ.get('SAFE_FUZZ_SECONDS', '15'))
    first = int(os.environ.get('SAFE_FUZZ_FIRST', '0'))
    t0, case = time.time(), first
    while time.time() - t0 < budget:
        rng = np.random.default_rng(700000 + case)
        n = int(rng.choice([rng.integers(2, 40), rng.integers(40, 400), rng.integers(400, 1500)]))
        m = int(rng.choice([1, rng.integers(2, 12), rng.integers(12, 150)]))
        e = max(1, int(n * float(rng.choice([0.4, 1.0, 2.5, 6.0])) / 2))
        eu, ev = rng.integers(0, n, size=e), rng.integers(0, n, size=e)          # (self loops and repeated edges included)
        dens = np.exp(rng.uniform(np.log(0.01), np.log(1.0), size=m))
        nb = (rng.uniform(size=(n, m)) < dens[None, :]).astype(np.float64)
        if m > 3:
            nb[:, 1] = 0
            nb[:, 2] = 1
            nb[:, 3] = 0
        tag = 'case %d: n=%d m=%d edges=%d' % (case, n, m, e)
        labels = be.enriched_components(ctx, n, eu, ev, nb)
        want = orc.top_attributes(nb, np.full(m, n), n, eu, ev, min_size=1)
        for j in range(m):
            lab = labels[j]
            assert np.array_equal(lab >= 0, nb[:, j] > 0), (tag, j)
            sizes = np.sort(np.bincount(lab[lab >= 0]))[::-1] if (lab >= 0).any() else np.zeros(0, int)
            sizes = sizes[sizes > 0]
            ws = want['size_connected_components'][j]
            assert np.array_equal(sizes, ws if ws is not None else np.zeros(0, int)), (tag, j)
            roots = np.unique(lab[lab >= 0])
            assert all(lab[r] == r for r in roots), (tag, j)
            assert len(roots) == want['num_connected_components'][j], (tag, j)
        if m > 1:
            with np.errstate(invalid='ignore', divide='ignore'):
                assert np.array_equal(be.jaccard_condensed(ctx, nb.T), pdist(nb.T, 'jaccard'), equal_nan=True), tag
        case += 1
    print('cases %d..%d' % (first, case - 1))
    assert case - first >= 20
