"""Full BASELINE.json sizes on the GPU, checked through size-independent properties and against the
oracle on a sample (the oracle needs ~0.4-1.4 s per permutation at this size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def cfg2():
    import safepy_amd
    from safepy_amd import workloads
    assert safepy_amd.device_count() >= 1
    data = workloads.costanzo_surrogate(seed=0)
    sf = safepy_amd.SAFE(verbose=False)
    sf.random_seed = 0
    sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
    sf.define_neighborhoods()                    # default metric, r = 0.1
    sf.load_attributes(attribute_file=data['attributes'])
    return safepy_amd, sf, data


def test_config2_neighborhood_statistics_and_symmetry(cfg2):
    amd, sf, data = cfg2
    a = sf.neighborhoods
    assert a.shape == (3971, 3971) and a.dtype == np.int64
    assert np.array_equal(a, a.T) and np.all(np.diag(a) == 1)
    counts = a.sum(axis=1)
    # surrogate tuned to the reference's known answer 37.5 +/- 56.7 (tests/test_neighborhoods.py:25-26)
    assert abs(counts.mean() - 37.5) < 5 and abs(counts.std() - 56.7) < 8
    # every member is within the weighted radius in the reference's own distance dict
    nd = sf.node_distances
    cutoff = 0.1 * (data['xy'][:, 0].max() - data['xy'][:, 0].min())
    assert all(d <= cutoff for row in list(nd.values())[:50] for d in row.values())


def test_config2_full_permutation_test_properties_and_sample(cfg2):
    amd, sf, data = cfg2
    b = data['attributes']
    n, m = b.shape
    nperm = 1000
    sf.compute_pvalues(how='randomization', num_permutations=nperm, verbose=False)
    assert sf.nes.shape == (n, m)
    a = sf.neighborhoods
    # (1) observed scores: exact integer neighbourhood counts (checked on a column sample with the oracle)
    cols = np.r_[0:8, m - 8:m, 2000:2008]
    want_ns = orc.compute_neighborhood_score(a, b[:, cols].astype(np.float64), 'sum')
    assert np.array_equal(sf.ns[:, cols], want_ns)
    # (2) p-values are multiples of 1/P in [0,1]; every permutation is <=, >= or both
    cn, cp = sf.pvalues_neg * nperm, sf.pvalues_pos * nperm
    assert np.array_equal(cn, np.round(cn)) and np.array_equal(cp, np.round(cp))
    assert cn.min() >= 0 and cn.max() <= nperm and cp.min() >= 0 and cp.max() <= nperm
    assert np.all(cn + cp >= nperm)                              # ties count on both sides (safe_extras.py:65-66)
    # (3) an attribute nobody carries / an empty observed count can never be beaten from below
    zero_obs = sf.ns == 0
    assert np.all(cp[zero_obs] == nperm)
    # (4) NES and binarisation follow from the p-values exactly as safe.py:546-554, 468-472
    with np.errstate(divide='ignore'):
        nes = -np.log10(np.where(sf.pvalues_pos == 0, 1 / nperm, sf.pvalues_pos)) + \
            np.log10(np.where(sf.pvalues_neg == 0, 1 / nperm, sf.pvalues_neg))
    assert np.array_equal(sf.nes, nes)
    assert np.array_equal(sf.nes_binary, (np.abs(nes) > -np.log10(0.05)).astype(np.float64))
    assert np.array_equal(sf.attributes['num_neighborhoods_enriched'].values, sf.nes_binary.sum(axis=0))
    # (5) the first permutations, replayed by the oracle on the sampled columns, give the same counts
    few = 12
    sf2 = amd.SAFE(verbose=False)
    sf2.random_seed = 0
    sf2.neighborhoods = a
    sf2.load_attributes(attribute_file=np.asfortranarray(b[:, cols]))
    # same row set moves as in the full matrix: all-NaN rows are all-NaN in every column of the surrogate
    sf2.compute_pvalues(how='randomization', num_permutations=few, verbose=False)
    want = orc.compute_pvalues(a, b[:, cols].astype(np.float64), enrichment_type='randomization',
                               num_permutations=few, random_seed=0)
    assert np.array_equal(sf2.pvalues_neg, want['pvalues_neg']) and np.array_equal(sf2.pvalues_pos, want['pvalues_pos'])


def test_config2_kernel_forms_agree_at_full_size(cfg2, monkeypatch):
    """Bit-sliced, scatter and f64 gather forms on a 256-attribute block x 64 permutations."""
    amd, sf, data = cfg2
    from safepy_amd import backend as be
    ctx = amd.Context.default(0)
    b = np.asfortranarray(data['attributes'][:, 1000:1256])
    n, m = b.shape
    out = {}
    for path in ('bits', 'scatter', 'gather'):
        monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
        attr = be.Attributes.from_host(ctx, b)
        perms = be.Permutations(ctx, n, attr.row_flags(), 64, 9)
        neg, pos = ctx.alloc_f64(n, m), ctx.alloc_f64(n, m)
        be.permtest_counts(ctx, sf._device_neighborhoods(), attr, perms, 'sum', None, neg.ptr, pos.ptr)
        assert ctx.last_kernel()[0].startswith('k_permtest_' + path)
        out[path] = (neg.download((n, m)), pos.download((n, m)))
        perms.close()
        attr.close()
    for path in ('scatter', 'gather'):
        assert np.array_equal(out[path][0], out['bits'][0]) and np.array_equal(out[path][1], out['bits'][1])


@pytest.mark.parametrize('m', [512, 3001])
def test_config4_shape_hypergeometric_properties(m):
    """20 000 nodes, euclidean r = 0.1, binary attributes (a column block of config 4; 3001 is neither a multiple
    of the 192-column groups nor of 4): split matrix-core form, sampled rows against SciPy, whole-matrix
    properties (NES = -log10 p, binarisation, per-attribute counts)."""
    import safepy_amd
    from safepy_amd import backend as be
    rng = np.random.default_rng(12)
    n = 20000
    xy = rng.uniform(size=(n, 2))
    b = (rng.uniform(size=(n, m)) < 0.01).astype(np.float32)
    b[rng.choice(n, 1000, replace=False)] = np.nan
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(xy)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues()                                        # 'auto' -> hypergeometric
    assert be.Context.default(0).last_kernel()[0] == 'k_hyp_emit'
    assert sf.pvalues_neg is None
    p = sf.pvalues_pos
    assert p.shape == (n, m) and np.all((p >= 0) & (p <= 1))
    with np.errstate(divide='ignore'):
        np.testing.assert_allclose(sf.nes, -np.log10(p), rtol=1e-12, atol=1e-12)          # safe.py:608
    assert np.array_equal(sf.nes_binary, (sf.nes > -np.log10(0.05)).astype(np.float64)) or \
        np.abs(sf.nes_binary - (sf.nes > -np.log10(0.05))).sum() <= 1e-6 * n * m            # (ties at the threshold: decided on p)
    assert np.array_equal(sf.nes_binary.sum(axis=0), sf.attributes['num_neighborhoods_enriched'].values)
    # rows sampled against scipy through the oracle
    rows = rng.choice(n, 40, replace=False)
    a_rows = np.zeros((40, n), dtype=np.int64)
    d = np.sqrt(((xy[rows, None, :] - xy[None, :, :]) ** 2).sum(-1))
    nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
    a_rows[d < nr] = 1
    notnan = ~np.isnan(b).all(axis=1)
    from scipy.stats import hypergeom
    hits = a_rows @ np.nan_to_num(b.astype(np.float64))
    size = a_rows @ notnan.astype(np.int64)
    want = hypergeom.sf(hits - 1, notnan.sum(), np.nansum(b, axis=0)[None, :], size[:, None])
    np.testing.assert_allclose(p[rows], want, rtol=1e-6, atol=1e-300)
    assert np.array_equal(sf.nes_binary[rows], (-np.log10(want) > -np.log10(0.05)).astype(np.float64))


def test_config5_shape_quantitative_permutation_test_sampled_rows():
    """20 000 nodes, euclidean r = 0.1 (578 members per neighborhood), quantitative f64 attributes
    with NaN rows and scattered NaNs (a 96-column block of config 5's per-rank share) through the
    matrix-core kernel.  Checked on sampled neighborhoods against a direct NumPy evaluation that
    uses the device's own permutation tables (themselves pinned to NumPy's stream elsewhere):
    every <= / >= count identical, observed scores to 1e-9."""
    import safepy_amd
    from safepy_amd import backend as be, workloads
    ctx = safepy_amd.Context.default(0)
    n, m, nperm, seed = 20000, 96, 40, 5
    xy = workloads.uniform_layout(4, n)
    b = workloads.quantitative_attributes(7, n, m)
    nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
    nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
    attr = be.Attributes.from_host(ctx, b)
    flags = attr.row_flags()
    assert flags.sum() == (~np.isnan(b)).any(axis=1).sum()
    perms = be.Permutations(ctx, n, flags, nperm, seed)
    ns, neg, pos = (ctx.alloc_f64(n, m) for _ in range(3))
    be.permtest_counts(ctx, nbr, attr, perms, 'sum', ns.ptr, neg.ptr, pos.ptr)
    assert ctx.last_kernel()[0] == 'k_permtest_mfma'
    ns, neg, pos = ns.download((n, m)), neg.download((n, m)), pos.download((n, m))
    table = perms.read()                                         # cur[p][i]: permuted matrix p = B[cur[p]]
    assert np.array_equal(np.sort(table[0]), np.arange(n))
    # size-independent properties over the whole block
    assert np.all((neg >= 0) & (neg <= nperm) & (pos >= 0) & (pos <= nperm)) and np.all(neg + pos >= nperm)
    # sampled rows, exact
    rng = np.random.default_rng(0)
    rows = rng.choice(n, 24, replace=False)
    rp, col = nbr.csr()
    b0 = np.nan_to_num(b)
    for i in rows:
        members = col[rp[i]:rp[i + 1]]
        d = np.sqrt(((xy[members] - xy[i]) ** 2).sum(axis=1))
        assert np.all(d < nr) and len(members) == (np.sqrt(((xy - xy[i]) ** 2).sum(axis=1)) < nr).sum()
        obs = b0[members].sum(axis=0)
        np.testing.assert_allclose(ns[i], obs, rtol=1e-9, atol=1e-9)
        s = b0[table[:, members]].sum(axis=1)                    # [P, m]
        # a comparison decided by less than the f64 rounding of a 600-term sum is not checkable from here
        clear = np.abs(s - obs) > 1e-9
        assert clear.mean() > 0.999
        le = ((s <= obs) & clear).sum(axis=0)
        ge = ((s >= obs) & clear).sum(axis=0)
        unclear = (~clear).sum(axis=0)
        assert np.all(np.abs(neg[i] - le) <= unclear) and np.all(np.abs(pos[i] - ge) <= unclear)
        assert np.array_equal(neg[i][unclear == 0], le[unclear == 0]) and np.array_equal(pos[i][unclear == 0], ge[unclear == 0])
    perms.close()
    attr.close()
    nbr.close()


def test_config5_shape_zscore_permutation_test_sampled_rows():
    """The same shape with neighborhood_score_type='z-score' (safe_extras.py:19-31): the filtered z-score kernel of round 6
    (k_permtest_mfma_gz: four of seven slices on the matrix cores, single-precision decision, the rest settled exactly by
    k_mfma_resolve_z) at configs[4]'s network.  Sampled neighborhoods against the oracle's score function applied to the
    matrices permuted with the device's own tables: observed z-scores to 1e-9, every <= / >= count identical wherever the
    compare is not decided by the f64 rounding of a 600-term sum; NaN scores compare false on both sides."""
    import safepy_amd
    from safepy_amd import backend as be, workloads
    ctx = safepy_amd.Context.default(0)
    n, m, nperm, seed = 20000, 80, 30, 11
    xy = workloads.uniform_layout(4, n)
    b = workloads.quantitative_attributes(9, n, m)
    nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
    nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
    attr = be.Attributes.from_host(ctx, b)
    flags = attr.row_flags()
    perms = be.Permutations(ctx, n, flags, nperm, seed)
    ns, neg, pos = (ctx.alloc_f64(n, m) for _ in range(3))
    be.permtest_counts(ctx, nbr, attr, perms, 'z-score', ns.ptr, neg.ptr, pos.ptr)
    assert ctx.last_kernel()[0] == 'k_permtest_mfma'
    core, undecided = be.last_mfma_filter(ctx)
    assert core == 4 and undecided >= 0                          # the filtered form ran to the end
    ns, neg, pos = ns.download((n, m)), neg.download((n, m)), pos.download((n, m))
    table = perms.read()
    rng = np.random.default_rng(1)
    rows = rng.choice(n, 20, replace=False)
    rp, col = nbr.csr()
    a_rows = np.zeros((len(rows), n), dtype=np.int64)
    for k, i in enumerate(rows):
        a_rows[k, col[rp[i]:rp[i + 1]]] = 1
    with np.errstate(invalid='ignore', divide='ignore'):
        obs = orc.compute_neighborhood_score(a_rows, b, 'z-score')
        np.testing.assert_allclose(ns[rows], obs, rtol=1e-9, atol=1e-9, equal_nan=True)
        le = np.zeros_like(obs)
        ge = np.zeros_like(obs)
        unclear = np.zeros_like(obs)
        for p_ in range(nperm):
            z = orc.compute_neighborhood_score(a_rows, b[table[p_]], 'z-score')
            clear = ~(np.abs(z - obs) <= 1e-9 * np.maximum(1.0, np.abs(obs)))        # (NaN on either side: clear, and false both ways)
            le += (z <= obs) & clear
            ge += (z >= obs) & clear
            unclear += ~clear
    assert unclear.mean() < 1e-3
    assert np.all(np.abs(neg[rows] - le) <= unclear) and np.all(np.abs(pos[rows] - ge) <= unclear)
    ok = unclear == 0
    assert np.array_equal(neg[rows][ok], le[ok]) and np.array_equal(pos[rows][ok], ge[ok])
    perms.close()
    attr.close()
    nbr.close()


@pytest.mark.parametrize('n,pre,expect', [(8300, '', 'k_permtest_bits_pre'), (8300, '0', 'k_permtest_bits'), (13000, '0', 'k_permtest_bits'),
                                          (20000, '', 'k_permtest_bits_pre'), (20477, '', 'k_permtest_bits_pre'),
                                          (20478, '', 'k_permtest_bits_pre'), (26001, '', 'k_permtest_bits_pre'), (32767, '', 'k_permtest_bits_pre'),
                                          (32768, '', 'k_permtest_mfma'), (20000, '0', 'k_permtest_mfma')])
def test_binary_randomization_beyond_the_16_bit_address_range(n, pre, expect, monkeypatch):
    """0/1 attributes under how='randomization' on networks too large for the blocked bit-sliced kernel (member ids as
    16-bit LDS addresses need 8 (N + 1) < 65536).  Up to N = 20 477 -- the word column of 8 bytes per node still fits a CU's
    LDS -- the pre-permuted form runs with SIXTEEN-wave workgroups and doubled ids in its lists (k_permtest_bits_pre<8, 16, 2>,
    round 6); SAFE_HIP_BITS_PRE=0 gives what ran before: the bit-sliced kernel with the permutation row staged in LDS (now also
    sixteen waves per workgroup; up to N ~ 13 600) and beyond that the matrix-core kernel in its exact two-slice regime.  From
    N = 20 478 to 32 767 (the resident lists hold 2 * id in 16 bits) the word column is kept as 32-attribute HALF words
    (k_permtest_bits_pre32: every word group in two passes); at 32 768 the matrix cores take over.  Sampled neighborhoods against a direct NumPy evaluation over the device's own permutation
    tables: integer sums, every count identical."""
    if pre:
        monkeypatch.setenv('SAFE_HIP_BITS_PRE', pre)
    import safepy_amd
    from safepy_amd import backend as be, workloads
    ctx = safepy_amd.Context.default(0)
    m, nperm, seed = 70, 24, 3
    xy = workloads.uniform_layout(9, n)
    rng = np.random.default_rng(n)
    b = (rng.uniform(size=(n, m)) < np.linspace(0.002, 0.3, m)).astype(np.float32)
    b[rng.choice(n, n // 50, replace=False)] = np.nan
    nr = 0.03 * (xy[:, 0].max() - xy[:, 0].min())
    nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
    attr = be.Attributes.from_host(ctx, b)
    perms = be.Permutations(ctx, n, attr.row_flags(), nperm, seed)
    ns, neg, pos = (ctx.alloc_f64(n, m) for _ in range(3))
    be.permtest_counts(ctx, nbr, attr, perms, 'sum', ns.ptr, neg.ptr, pos.ptr)
    assert ctx.last_kernel()[0] == expect
    ns, neg, pos = ns.download((n, m)), neg.download((n, m)), pos.download((n, m))
    table = perms.read()
    assert np.all(neg + pos >= nperm) and np.all((neg <= nperm) & (pos <= nperm))
    rp, col = nbr.csr()
    b0 = np.nan_to_num(b).astype(np.float64)
    for i in np.random.default_rng(1).choice(n, 40, replace=False):
        members = col[rp[i]:rp[i + 1]]
        obs = b0[members].sum(axis=0)
        assert np.array_equal(ns[i], obs)
        s = b0[table[:, members]].sum(axis=1)                    # [P, m], whole numbers
        assert np.array_equal(neg[i], (s <= obs).sum(axis=0)) and np.array_equal(pos[i], (s >= obs).sum(axis=0))
    perms.close()
    attr.close()
    nbr.close()


@pytest.mark.parametrize('n', [9000, 21000])
def test_fused_randomization_call_on_the_sixteen_wave_forms_against_the_oracle(n):
    """The whole fused call (safe_randomization: scores, both empirical p-value matrices, NES, nes_binary, enriched counts) on
    the networks of round 6's sixteen-wave bit-sliced forms -- N = 9000 full words, N = 21 000 half words -- against the oracle
    (safe.py:496-554, 468-472) with the SEEDED stream, hub neighborhoods above 1023 members included.  The oracle's dot products
    run in f32 on a dense 0/1 membership (sums of 0/1 below 2^24: exact), 12 permutations."""
    import safepy_amd
    from safepy_amd import backend as be, workloads
    ctx = safepy_amd.Context.default(0)
    m, nperm, seed = 70, 12, 17
    xy = workloads.uniform_layout(3, n)
    xy[:1200] = xy[0] + 0.004 * np.random.default_rng(2).normal(size=(1200, 2))             # a dense cluster: neighborhoods of 1024+ members
    rng = np.random.default_rng(n)
    b = (rng.uniform(size=(n, m)) < np.linspace(0.002, 0.5, m)).astype(np.float32)
    b[rng.choice(n, n // 40, replace=False)] = np.nan
    nr = 0.012 * (xy[:, 0].max() - xy[:, 0].min())
    nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
    counts = nbr.row_counts()
    assert 1024 <= counts.max() < 2048, counts.max()
    attr = be.Attributes.from_host(ctx, b)
    perms = be.Permutations(ctx, n, attr.row_flags(), nperm, seed)
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs])
    assert ctx.last_kernel()[0] == 'k_permtest_bits_pre'
    got = [o.download((n, m)) for o in outs[:5]] + [outs[5].download((m,))]
    rp, col = nbr.csr()
    a = np.zeros((n, n), dtype=np.float32)
    a[np.repeat(np.arange(n), np.diff(rp)), col] = 1
    want = orc.pvalues_by_randomization(a, b.copy(), 'sum', nperm, seed, 'both')
    nes_binary, num_enriched = orc.binarize(want['nes'], 0.05)
    for have, key in zip(got[:4], ('ns', 'pvalues_neg', 'pvalues_pos', 'nes')):
        np.testing.assert_array_equal(have, want[key], err_msg=key)
    np.testing.assert_array_equal(got[4], nes_binary)
    np.testing.assert_array_equal(got[5], num_enriched)
    for o in outs:
        o.free()
    perms.close()
    attr.close()
    nbr.close()
