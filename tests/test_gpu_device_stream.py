"""Unseeded runs (random_seed=None, the reference's default: OS entropy, safe.py:88 / safe_extras.py:46) generate their
permutation tables ON THE DEVICE (safe_perms_create_device).  There is no reference stream to match, so the tests pin
(1) the product's documented algorithm bit for bit against its restatement in the oracle, (2) what the reference's semantics
require of ANY stream -- every table row a uniform, independent permutation of the rows that hold a value, the others fixed
-- statistically, and (3) that the enrichment kernels consume such tables exactly like the NumPy-compatible ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


@pytest.fixture(scope='module')
def ctx(amd):
    return amd.Context.default(0)


@pytest.mark.parametrize('n,k_fixed,nperm', [(300, 17, 12), (9, 0, 40), (5, 4, 6), (4, 4, 3), (3971, 182, 3), (66000, 700, 2)])
def test_tables_equal_the_restated_algorithm(amd, ctx, n, k_fixed, nperm):
    from safepy_amd import backend as be
    rng = np.random.default_rng(n)
    flags = np.ones(n, dtype=np.uint8)
    flags[rng.choice(n, k_fixed, replace=False)] = 0
    key = int(rng.integers(0, 2 ** 63))
    perms = be.Permutations(ctx, n, flags, nperm, None, device_key=key)
    assert perms.timing()['role'] == 'device' and perms.device_key == key
    got = perms.read().astype(np.int64)
    perms.close()
    np.testing.assert_array_equal(got, orc.device_stream_tables(n, flags, nperm, key))
    assert (np.sort(got, axis=1) == np.arange(n)).all()                       # every row a permutation of 0 .. n-1
    assert (got[:, flags == 0] == np.flatnonzero(flags == 0)).all()            # rows without a value never move
    again = be.Permutations(ctx, n, flags, nperm, None, device_key=key)        # (buffers of the last handle are reused)
    np.testing.assert_array_equal(again.read().astype(np.int64), got)
    again.close()
    other = be.Permutations(ctx, n, flags, nperm, None)                        # a fresh entropy key: another stream
    assert other.device_key != key and (n < 6 or not np.array_equal(other.read(), got))
    other.close()


def test_uniform_and_independent(amd, ctx):
    """All 24 orders of 4 movable rows equally often; every (position, row) pair of a 50-row shuffle equally often;
    consecutive table rows unrelated -- chi-square against the uniform law at the 1e-4 level."""
    from scipy.stats import chi2
    from safepy_amd import backend as be
    flags = np.array([1, 0, 1, 1, 0, 1], dtype=np.uint8)
    P = 48000
    perms = be.Permutations(ctx, 6, flags, P, None, device_key=20240501)
    t = perms.read()[:, flags == 1]
    perms.close()
    codes = np.unique(t, axis=0, return_counts=True)[1]
    assert len(codes) == 24
    stat = ((codes - P / 24.0) ** 2 / (P / 24.0)).sum()
    assert stat < chi2.ppf(1 - 1e-4, 23), stat
    # consecutive rows: the pair (first entry of row q, first entry of row q + 1) is uniform on 4 x 4
    pair = np.zeros((6, 6))
    np.add.at(pair, (t[:-1, 0], t[1:, 0]), 1)
    pair = pair[np.ix_([0, 2, 3, 5], [0, 2, 3, 5])]
    stat = ((pair - (P - 1) / 16.0) ** 2 / ((P - 1) / 16.0)).sum()
    assert stat < chi2.ppf(1 - 1e-4, 15), stat
    n, P = 50, 40000
    perms = be.Permutations(ctx, n, np.ones(n, dtype=np.uint8), P, None, device_key=7)
    t = perms.read()
    perms.close()
    cells = np.zeros((n, n))
    np.add.at(cells, (np.tile(np.arange(n), P), t.reshape(-1)), 1)
    stat = ((cells - P / n) ** 2 / (P / n)).sum()
    assert stat < chi2.ppf(1 - 1e-4, (n - 1) ** 2), stat


@pytest.mark.parametrize('kind', ['binary', 'quantitative', 'z-score'])
def test_kernels_consume_device_tables_exactly(amd, ctx, kind):
    """The counts of a permutation test run on device-generated tables == a direct NumPy evaluation of those same tables
    (read back from the device): every kernel family takes them like any other table."""
    from safepy_amd import backend as be
    rng = np.random.default_rng(3)
    n, m, nperm = 500, 70, 33
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.1)
    b = (rng.uniform(size=(n, m)) < 0.06).astype(np.float64) if kind == 'binary' else np.round(rng.normal(size=(n, m)) * 16) / 16
    b[rng.choice(n, 20, replace=False)] = np.nan
    score = 'z-score' if kind == 'z-score' else 'sum'
    nbr = amd.Neighborhoods.from_dense(ctx, a)
    attr = be.Attributes.from_host(ctx, b)
    perms = be.Permutations(ctx, n, attr.row_flags(), nperm, None, device_key=99)
    tables = perms.read()
    ns, neg, pos = (ctx.alloc_f64(n, m) for _ in range(3))
    be.permtest_counts(ctx, nbr, attr, perms, score, ns.ptr, neg.ptr, pos.ptr)
    cn, cp = neg.download((n, m)), pos.download((n, m))
    obs = orc.compute_neighborhood_score(a, b, score)
    want_n, want_p = np.zeros((n, m)), np.zeros((n, m))
    with np.errstate(invalid='ignore'):
        for row in tables:
            sc = orc.compute_neighborhood_score(a, b[row], score)
            want_n += sc <= obs
            want_p += sc >= obs
    assert np.array_equal(cn, want_n) and np.array_equal(cp, want_p)
    for h in (perms, attr, nbr):
        h.close()


def test_unseeded_compute_pvalues(amd, monkeypatch):
    """SAFE.compute_pvalues with random_seed=None: device stream; repeatable with device_stream_key, different without;
    statistically the seeded run's p-values; SAFE_HIP_DEVICE_STREAM=0 falls back to the NumPy-compatible stream."""
    rng = np.random.default_rng(8)
    n, m, nperm = 600, 40, 400
    xy = rng.uniform(size=(n, 2))
    b = rng.normal(size=(n, m))
    b[rng.choice(n, 30, replace=False)] = np.nan

    def run(seed, key=None):
        sf = amd.SAFE(verbose=False)
        sf.graph = amd.LayoutGraph(xy)
        sf.random_seed = seed
        sf.device_stream_key = key
        sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
        sf.load_attributes(attribute_file=b.copy())
        sf.compute_pvalues(num_permutations=nperm, verbose=False)
        return sf.ns, sf.pvalues_pos, sf.pvalues_neg

    ns_a, p_a, q_a = run(None, key=5)
    ns_b, p_b, q_b = run(None, key=5)
    ns_c, p_c, _ = run(None)
    ns_s, p_s, q_s = run(123)
    assert np.array_equal(p_a, p_b) and np.array_equal(q_a, q_b)
    assert not np.array_equal(p_a, p_c)
    for ns in (ns_b, ns_c, ns_s):
        np.testing.assert_array_equal(ns, ns_a)                            # the observed scores do not depend on the stream
    # two independent estimates of the same p: their difference has variance 2 p (1 - p) / P (p from the pooled estimate)
    ok = ~np.isnan(p_a)
    pooled = 0.5 * (p_a[ok] + p_s[ok])
    z = (p_a[ok] - p_s[ok]) / np.sqrt((pooled * (1 - pooled) + 0.5 / nperm) * 2.0 / nperm)
    assert abs(z.mean()) < 0.05 and 0.8 < z.std() < 1.1 and np.abs(z).max() < 6, (z.mean(), z.std(), np.abs(z).max())
    assert np.allclose(p_a[ok] + q_a[ok], 1.0 + (p_a[ok] + q_a[ok] - 1.0).clip(0, None))     # ties only ever add to p + q
    monkeypatch.setenv('SAFE_HIP_DEVICE_STREAM', '0')
    from safepy_amd import backend as be
    perms = be.Permutations(amd.Context.default(0), 50, np.ones(50, dtype=np.uint8), 5, None)
    assert perms.timing()['role'] == 'own' and perms.device_key is None
    perms.close()
