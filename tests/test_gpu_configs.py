"""Every BASELINE.json configuration at ITS OWN size through the product path, checked against the oracle
(exactly where one column / a sample is cheap on the CPU) and through size-independent properties.

  configs[0]  Costanzo-shaped network, ONE quantitative attribute (1315 NaN rows, one zero), 1000 permutations;
              plus the one-column binary attribute through the hypergeometric path
  configs[2]  3971 x 4373 binary x 10 000 permutations (16-bit packed counters, multi-span launch plan)
  configs[3]  20 000 x 10 000 binary, hypergeometric, the whole call
  configs[4]  one rank's share: 20 000 x 6250 quantitative x 1000 permutations (matrix-core kernel)

(configs[1] is tests/test_gpu_fullsize.py.)  Also: compute_node_distances() for both shortest-path metrics
against the reference vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def costanzo():
    import safepy_amd
    from safepy_amd import workloads
    assert safepy_amd.device_count() >= 1
    data = workloads.costanzo_surrogate(seed=0)
    sf = safepy_amd.SAFE(verbose=False)
    sf.random_seed = 0
    sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
    sf.define_neighborhoods()                    # default metric (shortpath_weighted_layout), r = 0.1
    return safepy_amd, sf, data


def doxorubicin_like(n, seed=3):
    """The shape of the reference's single-attribute example (tests/test_enrichments.py:60-101 of the reference):
    one quantitative column, 1315 NaN, exactly one zero, about as many positives as negatives."""
    rng = np.random.default_rng(seed)
    col = rng.normal(size=n)
    col[rng.choice(n, 1315, replace=False)] = np.nan
    live = np.flatnonzero(~np.isnan(col))
    col[live[0]] = 0.0
    return col.reshape(n, 1)


# ------------------------------------------------------------------------------- configs[0] ----
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_config0_single_quantitative_attribute_1000_permutations_exact(costanzo, dtype):
    amd, sf, data = costanzo
    n = data['xy'].shape[0]
    b = doxorubicin_like(n).astype(dtype)
    assert np.isnan(b).sum() == 1315 and (b == 0).sum() == 1
    sf.load_attributes(attribute_file=b.copy())
    sf.compute_pvalues(num_permutations=1000)                  # 'auto' -> randomization (values outside {0, 1})
    assert sf.pvalues_neg is not None
    want = orc.compute_pvalues(sf.neighborhoods, b.copy(), num_permutations=1000, random_seed=0)
    np.testing.assert_allclose(sf.ns, want['ns'], rtol=1e-9, atol=1e-12)
    # empirical p-values are counts / 1000: equal unless a comparison is decided inside the rounding of a
    # 40-term f64 sum (none observed; a single flip would be 1e-3, far outside the north star's 1e-6)
    assert np.array_equal(sf.pvalues_neg, want['pvalues_neg'])
    assert np.array_equal(sf.pvalues_pos, want['pvalues_pos'])
    assert np.array_equal(sf.nes, want['nes'])
    assert np.array_equal(sf.nes_binary, want['nes_binary'])
    assert sf.attributes['num_neighborhoods_enriched'].values[0] == want['num_neighborhoods_enriched'][0]
    # the reference's known answer for its own data is 637 +/- 20 enriched neighborhoods of 3971
    # (tests/test_enrichments.py:98-101); the surrogate is not that data -- only sanity here
    assert 0 < sf.nes_binary.sum() < n


def test_config0_single_binary_attribute_hypergeometric(costanzo):
    amd, sf, data = costanzo
    n = data['xy'].shape[0]
    rng = np.random.default_rng(8)
    b = (rng.uniform(size=(n, 1)) < 0.03).astype(np.float64)
    b[rng.choice(n, 182, replace=False)] = np.nan
    sf.load_attributes(attribute_file=b.copy())
    sf.ns = sf.pvalues_neg = None
    sf.compute_pvalues()                                        # 'auto' -> hypergeometric
    assert sf.pvalues_neg is None and sf.ns is None            # untouched on this path (safe.py:556-608)
    want = orc.compute_pvalues(sf.neighborhoods, b.copy())
    np.testing.assert_allclose(sf.pvalues_pos, want['pvalues_pos'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-6, atol=1e-9)
    assert np.array_equal(sf.nes_binary, want['nes_binary'])


# ------------------------------------------------------------------------------- configs[2] ----
def test_config2_ten_thousand_permutations(costanzo):
    """3971 x 4373 x 10 000: whole-matrix properties, and on sampled (neighborhood, attribute) pairs the exact
    <= / >= counts over ALL 10 000 permutations, evaluated in NumPy from the device's own permutation tables
    (pinned to NumPy's legacy stream in tests/test_gpu_parity.py and tests/test_abi.py)."""
    amd, sf, data = costanzo
    from safepy_amd import backend as be
    b = data['attributes']
    n, m = b.shape
    nperm = 10000
    sf.random_seed = 0
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(how='randomization', num_permutations=nperm, verbose=False)
    ctx = amd.Context.default(0)
    assert ctx.last_kernel()[0].startswith('k_permtest_bits')
    cn, cp = np.round(sf.pvalues_neg * nperm), np.round(sf.pvalues_pos * nperm)
    assert np.array_equal(cn / nperm, sf.pvalues_neg) and np.array_equal(cp / nperm, sf.pvalues_pos)      # p = count / P (safe.py:532-533)
    assert cn.min() >= 0 and cn.max() <= nperm and cp.min() >= 0 and cp.max() <= nperm
    assert np.all(cn + cp >= nperm)                              # ties count on both sides (safe_extras.py:65-66)
    assert np.all(cp[sf.ns == 0] == nperm)
    nan_rows = np.isnan(b).all(axis=1)
    assert np.all(cn[:, :] <= nperm) and np.all(sf.ns[:, 0] >= 0) and nan_rows.sum() == 182
    with np.errstate(divide='ignore'):
        nes = -np.log10(np.where(sf.pvalues_pos == 0, 1 / nperm, sf.pvalues_pos)) + \
            np.log10(np.where(sf.pvalues_neg == 0, 1 / nperm, sf.pvalues_neg))
    assert np.array_equal(sf.nes, nes)
    assert np.array_equal(sf.nes_binary, (np.abs(nes) > -np.log10(0.05)).astype(np.float64))
    assert np.array_equal(sf.attributes['num_neighborhoods_enriched'].values, sf.nes_binary.sum(axis=0))
    # a p-value that is exactly 0 needs all 10 000 permutations on one side: resolution 1e-4 is really used
    assert np.unique(sf.pvalues_pos).size > 100

    # ---- sampled pairs, all 10 000 permutations, exact
    flags = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    perms = be.Permutations(ctx, n, flags, nperm, 0)
    table = perms.read()                                         # cur[p][i]: permuted matrix p = B[cur[p]]
    perms.close()
    rp, col = sf._device_neighborhoods().csr()
    rng = np.random.default_rng(1)
    counts = np.diff(rp)
    rows = np.r_[rng.choice(n, 20, replace=False), np.argsort(counts)[-4:], np.argsort(counts)[:2]]
    cols = np.r_[0:6, m - 6:m, rng.choice(m, 20, replace=False), np.argsort(np.nansum(b, axis=0))[-4:]]
    b0 = np.nan_to_num(b[:, cols].astype(np.float64))
    for i in rows:
        members = col[rp[i]:rp[i + 1]]
        obs = b0[members].sum(axis=0)
        assert np.array_equal(sf.ns[i, cols], obs)
        s = b0[table[:, members]].sum(axis=1)                    # [P, len(cols)] exact small integers
        assert np.array_equal(cn[i, cols], (s <= obs).sum(axis=0))
        assert np.array_equal(cp[i, cols], (s >= obs).sum(axis=0))


# ------------------------------------------------------------------------------- configs[3] ----
def test_config3_whole_call_20000_by_10000_hypergeometric():
    """The whole configs[3] call -- 20 000 nodes (euclidean r = 0.1), 10 000 binary attributes, 'auto' ->
    hypergeometric -- properties on the full matrices and sampled rows against SciPy."""
    import safepy_amd
    from safepy_amd import backend as be
    from scipy.stats import hypergeom
    rng = np.random.default_rng(4)
    n, m = 20000, 10000
    xy = rng.uniform(size=(n, 2))
    b = (rng.uniform(size=(n, m)) < 0.01).astype(np.float32)
    b[rng.choice(n, 1000, replace=False)] = np.nan               # the 5 % NaN-row variant of SURVEY 8(d)
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(xy)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues()
    assert be.Context.default(0).last_kernel()[0] == 'k_hyp_emit'
    assert sf.pvalues_neg is None
    p = sf.pvalues_pos
    assert p.shape == (n, m) and p.min() >= 0 and p.max() <= 1 and not np.isnan(p).any()
    nes, nb = sf.nes, sf.nes_binary
    thr = -np.log10(0.05)
    for r0 in range(0, n, 2500):                                 # in slabs: the temporaries of a 1.6 GB matrix are 1.6 GB each
        sl = slice(r0, r0 + 2500)
        with np.errstate(divide='ignore'):
            np.testing.assert_allclose(nes[sl], -np.log10(p[sl]), rtol=1e-12, atol=1e-12)      # safe.py:608
        assert np.abs(nb[sl] - (nes[sl] > thr)).sum() <= 1e-6 * 2500 * m          # (ties at the threshold are decided on p)
    assert np.array_equal(nb.sum(axis=0), sf.attributes['num_neighborhoods_enriched'].values)
    # a neighborhood with no annotated member: X = 0 -> sf(-1) = 1 exactly
    rows = rng.choice(n, 48, replace=False)
    nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
    d = np.sqrt(((xy[rows, None, :] - xy[None, :, :]) ** 2).sum(-1))
    a_rows = (d < nr).astype(np.int64)
    notnan = ~np.isnan(b).all(axis=1)
    b0 = np.nan_to_num(b)
    hits = a_rows.astype(np.float32) @ b0                        # exact: small integers
    size = a_rows @ notnan.astype(np.int64)
    k_col = b0.sum(axis=0, dtype=np.float64)
    want = hypergeom.sf(hits.astype(np.float64) - 1, notnan.sum(), k_col[None, :], size[:, None])
    np.testing.assert_allclose(p[rows], want, rtol=1e-6, atol=1e-300)
    assert np.array_equal(nb[rows], (-np.log10(want) > thr).astype(np.float64))
    assert np.all(p[rows][hits == 0] == 1.0)


# ------------------------------------------------------------------------------- configs[4] ----
def test_config4_rank_share_20000_by_6250_by_1000_matrix_core():
    """One rank's share of configs[4]: 20 000 nodes x 6250 quantitative f64 attributes x 1000 permutations through
    SAFE.compute_pvalues (matrix-core kernel).  Whole-block properties; sampled neighborhoods x sampled columns
    evaluated in NumPy over all 1000 permutations from the device's own tables."""
    import safepy_amd
    from safepy_amd import backend as be, workloads
    ctx = safepy_amd.Context.default(0)
    n, m, nperm = 20000, 6250, 1000
    xy = workloads.uniform_layout(4, n)
    b = workloads.quantitative_attributes(11, n, m)
    sf = safepy_amd.SAFE(verbose=False)
    sf.random_seed = 0
    sf.graph = safepy_amd.LayoutGraph(xy)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(num_permutations=nperm)
    assert ctx.last_kernel()[0] == 'k_permtest_mfma'
    cn, cp = np.round(sf.pvalues_neg * nperm), np.round(sf.pvalues_pos * nperm)
    assert np.array_equal(cn / nperm, sf.pvalues_neg) and np.array_equal(cp / nperm, sf.pvalues_pos)
    assert cn.min() >= 0 and cn.max() <= nperm and cp.min() >= 0 and cp.max() <= nperm
    assert np.all(cn + cp >= nperm)
    assert np.array_equal(sf.nes_binary.sum(axis=0), sf.attributes['num_neighborhoods_enriched'].values)
    with np.errstate(divide='ignore'):
        nes = -np.log10(np.where(sf.pvalues_pos == 0, 1 / nperm, sf.pvalues_pos)) + \
            np.log10(np.where(sf.pvalues_neg == 0, 1 / nperm, sf.pvalues_neg))
    assert np.array_equal(sf.nes, nes)
    del nes
    flags = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    perms = be.Permutations(ctx, n, flags, nperm, 0)
    table = perms.read()
    perms.close()
    rp, col = sf._device_neighborhoods().csr()
    rng = np.random.default_rng(0)
    rows = rng.choice(n, 10, replace=False)
    cols = np.r_[0:4, m - 4:m, rng.choice(m, 24, replace=False)]
    b0 = np.nan_to_num(b[:, cols])
    ns = sf.ns
    for i in rows:
        members = col[rp[i]:rp[i + 1]]
        obs = b0[members].sum(axis=0)
        np.testing.assert_allclose(ns[i, cols], obs, rtol=1e-9, atol=1e-9)
        s = b0[table[:, members]].sum(axis=1)                    # [P, len(cols)]
        clear = np.abs(s - obs) > 1e-9                           # (a comparison inside f64 rounding of a 600-term sum is not checkable here)
        assert clear.mean() > 0.999
        unclear = (~clear).sum(axis=0)
        le, ge = ((s <= obs) & clear).sum(axis=0), ((s >= obs) & clear).sum(axis=0)
        assert np.all(np.abs(cn[i, cols] - le) <= unclear) and np.all(np.abs(cp[i, cols] - ge) <= unclear)
        ok = unclear == 0
        assert np.array_equal(cn[i, cols][ok], le[ok]) and np.array_equal(cp[i, cols][ok], ge[ok])


# ---------------------------------------------------------------- compute_node_distances, shortest paths ----
def _dense(nd, n):
    got = np.full((n, n), np.inf)
    for s, row in nd.items():
        for t, dist in row.items():
            got[s, t] = dist
    return got


@pytest.mark.parametrize('radius', [0.08, 0.2])
def test_compute_node_distances_weighted_shortpath_vs_reference(golden_nbr, radius):
    """The additive method on its own (no define_neighborhoods call): the dict-of-dicts the reference stores at
    safe.py:417, bit-exact path lengths, and self.neighborhoods left untouched."""
    import safepy_amd
    g = golden_nbr
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(g['xy'], g['edge_u'], g['edge_v'], length=g['edge_length'])
    sf.compute_node_distances(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=radius)
    assert sf.neighborhoods is None
    want = g['swl_dist_r%g' % radius]
    nd = sf.node_distances
    assert isinstance(nd, dict) and len(nd) == want.shape[0]
    assert np.array_equal(_dense(nd, want.shape[0]), want)
    assert np.array_equal(np.isfinite(want), g['swl_r%g' % radius].astype(bool))


def test_compute_node_distances_unweighted_shortpath_vs_reference(golden_nbr):
    """'shortpath' (hop counts): the reference masks at radius 1, 2, 3 pin every distance up to 3."""
    import safepy_amd
    g = golden_nbr
    n = g['xy'].shape[0]
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(g['xy'], g['edge_u'], g['edge_v'])
    sf.compute_node_distances(node_distance_metric='shortpath', neighborhood_radius=3)
    r1, r2, r3 = (g['shortpath_r%d' % r].astype(bool) for r in (1, 2, 3))
    want = np.full((n, n), np.inf)
    want[r3] = 3
    want[r2] = 2
    want[r1] = 1
    want[np.eye(n, dtype=bool)] = 0
    assert np.array_equal(_dense(sf.node_distances, n), want)
