"""Random cases of the neighborhood definition against the oracle (safe.py:387-420): all-pairs Euclidean membership with
coincident points and pairs EXACTLY at the threshold, bounded shortest paths on random graphs (several components, isolated
nodes, self loops, zero-length edges, integer / dyadic weights with many equal-length paths, random real weights), unweighted
hop counts.  Memberships and path lengths are compared bit for bit.

SAFE_FUZZ_SECONDS (default 20) bounds the run; SAFE_FUZZ_FIRST names the first case."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


def _points(rng, n):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        xy = rng.uniform(-3, 7, size=(n, 2))
    elif kind == 1:                                     # an integer grid: many pairs at exactly the same distance
        xy = rng.integers(0, max(2, int(np.sqrt(n)) + 1), size=(n, 2)).astype(np.float64)
    elif kind == 2:                                     # clusters
        k = int(rng.integers(1, 6))
        xy = rng.uniform(size=(k, 2))[rng.integers(0, k, size=n)] + rng.normal(size=(n, 2)) * 0.02
    else:                                               # large offsets: the differences cancel
        xy = 1e6 + rng.uniform(size=(n, 2))
    if n > 10:
        xy[5] = xy[3]
    return xy


def _graph(rng, n):
    deg = float(rng.choice([0.6, 1.5, 3.0, 8.0]))
    e = max(1, int(n * deg / 2))
    u = rng.integers(0, n, size=e)
    v = rng.integers(0, n, size=e)
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    _, keep = np.unique(lo * n + hi, return_index=True)             # a simple graph: the reference's nx.Graph holds one edge per pair
    u, v = u[np.sort(keep)], v[np.sort(keep)]
    kind = str(rng.choice(['real', 'integer', 'dyadic', 'zeros']))
    if kind == 'real':
        w = rng.uniform(0.01, 1.0, size=u.shape[0])
    elif kind == 'integer':
        w = rng.integers(1, 4, size=u.shape[0]).astype(np.float64)
    elif kind == 'dyadic':
        w = rng.integers(1, 17, size=u.shape[0]) / 16.0
    else:
        w = rng.uniform(0.0, 1.0, size=u.shape[0]) * (rng.uniform(size=u.shape[0]) < 0.7)
    return u.astype(np.int32), v.astype(np.int32), w, kind


def test_random_neighborhood_definitions_against_the_oracle():
    import safepy_amd as amd
    assert amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    ctx = amd.Context.default(0)
    budget = float(os.environ.get('SAFE_FUZZ_SECONDS', '20'))
    first = int(os.environ.get('SAFE_FUZZ_FIRST', '0'))
    t0, case, seen = time.time(), first, {}
    while time.time() - t0 < budget:
        rng = np.random.default_rng(300000 + case)
        what = str(rng.choice(['euclidean', 'weighted', 'hops']))
        if what == 'euclidean':
            n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 600), rng.integers(600, 2600)]))
            xy = _points(rng, n)
            r = float(rng.choice([0.01, 0.05, 0.15, 0.5, 1.5]))
            if n > 10 and np.ptp(xy[:, 0]) > 0:                      # a pair exactly at the threshold (strict <: not a member)
                xy[7] = xy[3] + np.array([orc.layout_radius(xy[:, 0], r), 0.0])
            tag = 'case %d: euclidean n=%d radius=%g' % (case, n, r)
            want = orc.neighborhoods_euclidean(xy, r)
            nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], r))
            got = nbr.to_dense()
            assert np.array_equal(got, want), tag
            assert np.array_equal(nbr.row_counts(), want.sum(axis=1)), tag
        else:
            n = int(rng.choice([rng.integers(2, 30), rng.integers(30, 150), rng.integers(150, 400)]))
            eu, ev, ew, kind = _graph(rng, n)
            if what == 'hops':
                ew, kind = np.ones(eu.shape[0]), 'ones'
                cutoff = float(rng.integers(1, 5))
            else:
                cutoff = float(np.quantile(ew, rng.uniform(0.2, 1.0)) * rng.choice([0.5, 1.0, 2.0, 4.0]))
                if kind in ('integer', 'dyadic') and rng.uniform() < 0.5:
                    cutoff = float(np.round(cutoff * 16) / 16)      # paths of exactly the cutoff's length are kept (<=)
            tag = 'case %d: %s n=%d edges=%d weights=%s cutoff=%r' % (case, what, n, eu.shape[0], kind, cutoff)
            want, want_d = orc.neighborhoods_shortpath(n, eu, ev, ew, cutoff)
            nbr = amd.Neighborhoods.shortpath(ctx, n, eu, ev, None if what == 'hops' and rng.uniform() < 0.5 else ew, cutoff,
                                              keep_distances=True)
            assert np.array_equal(nbr.to_dense(), want), tag
            assert np.array_equal(nbr.distances(), want_d), tag
        nbr.close()
        seen[what] = seen.get(what, 0) + 1
        case += 1
    print('cases %d..%d: %s' % (first, case - 1, seen))
    assert case - first >= 20 and len(seen) == 3, seen
