"""Parity of the HIP path (through the C ABI) against the golden vectors produced by the
real reference and against the CPU oracle on seeded inputs.  Needs an MI355X."""
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


def _layout_graph(amd, g):
    return amd.LayoutGraph(g['xy'], g['edge_u'], g['edge_v'], length=g['edge_length'])


def _safe(amd, graph, **attrs):
    sf = amd.SAFE(verbose=False)
    sf.graph = graph
    for k, v in attrs.items():
        setattr(sf, k, v)
    return sf


# ------------------------------------------------------------------ neighborhoods ----

@pytest.mark.parametrize('radius', [0.05, 0.15])
def test_euclidean_mask_bit_exact_vs_reference(amd, golden_nbr, radius):
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=radius)
    got = sf.neighborhoods
    assert got.dtype == np.int64 and got.flags['C_CONTIGUOUS']
    assert np.array_equal(got, golden_nbr['euclidean_r%g' % radius].astype(np.int64))


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 1000, 2049])
def test_euclidean_mask_vs_oracle_sizes(amd, ctx, n):
    rng = np.random.default_rng(n)
    xy = rng.uniform(-3, 7, size=(n, 2))
    # duplicates and exact-threshold pairs
    if n > 10:
        xy[5] = xy[3]
        xy[7] = xy[3] + np.array([0.1 * 10.0, 0.0])
    want = orc.neighborhoods_euclidean(xy, 0.1)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    assert np.array_equal(nbr.to_dense(), want)
    assert np.array_equal(nbr.row_counts(), want.sum(axis=1))
    rp, col = nbr.csr()
    assert rp[-1] == want.sum() and np.array_equal(np.nonzero(want)[1], col)
    nbr.close()


@pytest.mark.parametrize('n', [257, 1000, 1001])
def test_fused_dense_kernel_mask_and_distances(amd, ctx, golden_nbr, n):
    if n == 257:
        xy = golden_nbr['xy']
        want_d = golden_nbr['euclidean_dist']
    else:
        xy = np.random.default_rng(n).normal(size=(n, 2))
        want_d = orc.euclidean_distances(xy)
    nr = orc.layout_radius(xy[:, 0], 0.15)
    d_xy = ctx.alloc(xy.nbytes)
    d_xy.upload(xy)
    d_mask = ctx.alloc(n * n * 8)
    d_dist = ctx.alloc(n * n * 8)
    ctx.euclidean_dense(d_xy.ptr, n, nr, d_mask.ptr, d_dist.ptr)
    mask = d_mask.download((n, n), np.int64)
    dist = d_dist.download((n, n), np.float64)
    assert np.array_equal(dist, want_d)                       # bit-exact f64 (no FMA, IEEE sqrt)
    assert np.array_equal(mask, (want_d < nr).astype(np.int64))


def test_compute_node_distances_euclidean(amd, golden_nbr):
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.compute_node_distances(node_distance_metric='euclidean')
    assert np.array_equal(sf.node_distances, golden_nbr['euclidean_dist'])
    assert sf.neighborhoods is None


def test_edge_lengths_bit_exact(amd, ctx, golden_nbr):
    g = golden_nbr
    assert np.array_equal(ctx.edge_lengths(g['xy'], g['edge_u'], g['edge_v']), g['edge_length'])


@pytest.mark.parametrize('radius', [0.08, 0.2])
def test_weighted_shortpath_vs_reference(amd, golden_nbr, radius):
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=radius)
    assert np.array_equal(sf.neighborhoods, golden_nbr['swl_r%g' % radius].astype(np.int64))
    want = golden_nbr['swl_dist_r%g' % radius]
    nd = sf.node_distances
    assert isinstance(nd, dict) and len(nd) == want.shape[0]
    got = np.full(want.shape, np.inf)
    for s, row in nd.items():
        for t, d in row.items():
            got[s, t] = d
    assert np.array_equal(got, want)                          # bit-exact path lengths


def test_shortpath_distances_outlive_their_device_handle(amd, golden_nbr):
    """node_distances of a shortest-path metric stay on the device until first read; they must survive the
    neighborhoods being redefined with the euclidean metric (which leaves node_distances alone, safe.py:389-399),
    a user-supplied membership, and pickling -- all before anybody looked at them."""
    import pickle
    want = golden_nbr['swl_dist_r0.2']

    def dense(nd):
        got = np.full(want.shape, np.inf)
        for s, row in nd.items():
            for t, d in row.items():
                got[s, t] = d
        return got

    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    assert sf.__dict__['_node_distances'][0] == 'device-shortpath'
    blob = pickle.dumps(sf)
    assert np.array_equal(dense(pickle.loads(blob).node_distances), want)
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
    assert np.array_equal(dense(sf.node_distances), want)
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    sf.neighborhoods = np.eye(want.shape[0], dtype=np.int64)
    assert np.array_equal(dense(sf.node_distances), want)


@pytest.mark.parametrize('radius', [1, 2, 3])
def test_unweighted_shortpath_vs_reference(amd, golden_nbr, radius):
    g = golden_nbr
    sf = _safe(amd, amd.LayoutGraph(g['xy'], g['edge_u'], g['edge_v']))
    sf.define_neighborhoods(node_distance_metric='shortpath', neighborhood_radius=radius)
    assert np.array_equal(sf.neighborhoods, g['shortpath_r%d' % radius].astype(np.int64))


def test_shortpath_with_networkx_graph(amd, golden_nbr):
    nx = pytest.importorskip('networkx')
    g = golden_nbr
    graph = nx.Graph()
    for i, (x, y) in enumerate(g['xy']):
        graph.add_node(i, x=float(x), y=float(y), label='n%d' % i, label_orf='ORF%d' % i)
    for u, v, w in zip(g['edge_u'], g['edge_v'], g['edge_length']):
        graph.add_edge(int(u), int(v), length=float(w))
    sf = _safe(amd, graph)
    sf.define_neighborhoods(neighborhood_radius=0.2)          # default metric
    assert sf.node_distance_metric == 'shortpath_weighted_layout'
    assert np.array_equal(sf.neighborhoods, g['swl_r0.2'].astype(np.int64))


def test_shortpath_disconnected_and_isolated(amd, ctx):
    # two components + an isolated node + a self loop + a zero-length edge
    eu = np.array([0, 1, 3, 4, 4, 2], dtype=np.int32)
    ev = np.array([1, 2, 4, 5, 4, 0], dtype=np.int32)
    ew = np.array([0.5, 0.25, 1.0, 0.0, 3.0, 0.875])
    want, want_d = orc.neighborhoods_shortpath(7, eu, ev, ew, 0.75)
    nbr = amd.Neighborhoods.shortpath(ctx, 7, eu, ev, ew, 0.75, keep_distances=True)
    assert np.array_equal(nbr.to_dense(), want)
    assert np.array_equal(nbr.distances(), want_d)


def test_dense_roundtrip_and_validation(amd, ctx):
    rng = np.random.default_rng(5)
    a = (rng.uniform(size=(130, 130)) < 0.1).astype(np.int64)
    nbr = amd.Neighborhoods.from_dense(ctx, a)
    assert np.array_equal(nbr.to_dense(), a) and nbr.nnz == a.sum()
    a[3, 4] = 2
    with pytest.raises(amd.SafeHipError):
        amd.Neighborhoods.from_dense(ctx, a)
    with pytest.raises(ValueError):
        amd.Neighborhoods.from_dense(ctx, np.zeros((3, 4)))


# ---------------------------------------------------------------------- RNG stream ----

def test_permutation_tables_vs_numpy_stream(amd, ctx, golden_enr):
    b = golden_enr['b_q']
    want = orc.permutation_index_table(b, 25, 29)
    movable = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    perms = amd.Permutations(ctx, b.shape[0], movable, 25, 29)
    assert np.array_equal(perms.read().astype(np.int64), want)
    perms.close()


def test_permutation_ranges_add_up_to_the_whole_call(amd, ctx, golden_enr):
    """safe_perms_slice + safe_outputs_from_counts (the permutation-axis split, safe.py:489-519): the counts of the ranges
    [0,7), [7,7), [7,19), [19,25) of one seeded stream add up to the counts of the 25-permutation call, every kernel family,
    and the outputs rebuilt from the summed counts are the outputs of the whole call."""
    from safepy_amd import backend as be
    from safepy_amd import sharding
    g = golden_enr
    a = g['A'].astype(np.int64)
    n = a.shape[0]
    nbr = amd.Neighborhoods.from_dense(ctx, a)
    for b, score in ((g['b_bin'], 'sum'), (g['b_q'], 'sum'), (g['b_q'], 'z-score')):
        b = np.ascontiguousarray(b, dtype=np.float64)
        m = b.shape[1]
        attr = be.Attributes.from_host(ctx, b)
        whole = be.Permutations(ctx, n, attr.row_flags(), 25, 31)
        ns, neg, pos = (ctx.alloc_f64(n, m) for _ in range(3))
        be.permtest_counts(ctx, nbr, attr, whole, score, ns.ptr, neg.ptr, pos.ptr)
        want_n, want_p, want_ns = neg.download((n, m)), pos.download((n, m)), ns.download((n, m))
        tables = whole.read()
        sum_n, sum_p = np.zeros((n, m)), np.zeros((n, m))
        for p0, p1 in ((0, 7), (7, 7), (7, 19), (19, 25)):
            part = whole.slice(p0, p1)
            assert part.count == p1 - p0 and np.array_equal(part.read(), tables[p0:p1])
            if p1 > p0:
                be.permtest_counts(ctx, nbr, attr, part, score, ns.ptr, neg.ptr, pos.ptr)
                sum_n += neg.download((n, m))
                sum_p += pos.download((n, m))
                assert np.array_equal(ns.download((n, m)), want_ns, equal_nan=True)
            part.close()
        assert np.array_equal(sum_n, want_n) and np.array_equal(sum_p, want_p)
        whole.close()
        # outputs from the summed counts == the whole call's outputs (one process, no process group)
        out = sharding.permutation_split_randomization(ctx, nbr, b, 25, 31, neighborhood_score_type=score)
        perms = be.Permutations(ctx, n, attr.row_flags(), 25, 31)
        bufs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
        be.randomization(ctx, nbr, attr, perms, score, 'both', 0.05, [x.ptr for x in bufs])
        ctx.sync()
        for key, buf in zip(('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'), bufs):
            assert np.array_equal(out[key], buf.download((n, m)), equal_nan=True), (score, key)
        assert np.array_equal(out['num_neighborhoods_enriched'], bufs[5].download((m,)))
        perms.close()
        attr.close()
    with pytest.raises(Exception):
        be.Permutations(ctx, n, np.ones(n, dtype=np.uint8), 5, 1).slice(3, 9)
    nbr.close()


def test_caller_supplied_permutation_tables(amd, ctx, golden_enr):
    """safe_perms_create_from_table: the `perm` of safe_extras.py:58 produced elsewhere.  (1) NumPy's own composed tables
    fed back in reproduce the seeded run; (2) tables from another generator give the counts a direct NumPy evaluation
    of those tables gives; (3) a row that is not a permutation is refused."""
    from safepy_amd import backend as be
    g = golden_enr
    a = g['A'].astype(np.int64)
    n = a.shape[0]
    nbr = amd.Neighborhoods.from_dense(ctx, a)
    for b in (g['b_bin'], g['b_q']):
        b = np.ascontiguousarray(b, dtype=np.float64)
        m = b.shape[1]
        attr = be.Attributes.from_host(ctx, b)
        for tables in (orc.permutation_index_table(b, 25, 31),
                       np.stack([np.random.default_rng(s).permutation(n) for s in range(17)])):
            perms = be.Permutations.from_table(ctx, tables)
            assert np.array_equal(perms.read().astype(np.int64), tables)
            neg, pos = ctx.alloc_f64(n, m), ctx.alloc_f64(n, m)
            be.permtest_counts(ctx, nbr, attr, perms, 'sum', None, neg.ptr, pos.ptr)
            cn, cp = neg.download((n, m)), pos.download((n, m))
            perms.close()
            b0 = np.nan_to_num(b)
            obs = a @ b0
            want_n, want_p = np.zeros((n, m)), np.zeros((n, m))
            for row in tables:
                sc = a @ b0[row]
                close = np.abs(sc - obs) <= 1e-9 * np.maximum(1.0, np.abs(obs))      # (real-valued sums: order of summation)
                want_n += (sc <= obs) | close
                want_p += (sc >= obs) | close
            exact = np.array_equal(b0, np.round(b0))
            if exact:
                assert np.array_equal(cn, want_n) and np.array_equal(cp, want_p)
            else:
                assert np.abs(cn - want_n).max() <= 1 and np.abs(cp - want_p).max() <= 1 and (cn != want_n).mean() < 1e-3
        attr.close()
    cn_seeded, cp_seeded = amd.run_permutations((a, g['b_bin'], 'sum', 25, 31), verbose=False)
    attr = be.Attributes.from_host(ctx, np.ascontiguousarray(g['b_bin'], dtype=np.float64))
    perms = be.Permutations.from_table(ctx, orc.permutation_index_table(g['b_bin'], 25, 31))
    m = g['b_bin'].shape[1]
    neg, pos = ctx.alloc_f64(n, m), ctx.alloc_f64(n, m)
    be.permtest_counts(ctx, nbr, attr, perms, 'sum', None, neg.ptr, pos.ptr)
    assert np.array_equal(neg.download((n, m)), cn_seeded) and np.array_equal(pos.download((n, m)), cp_seeded)
    perms.close()
    attr.close()
    bad = np.tile(np.arange(n, dtype=np.int32), (3, 1))
    bad[1, 5] = bad[1, 6]
    with pytest.raises(amd.SafeHipError):
        be.Permutations.from_table(ctx, bad)
    nbr.close()


def test_rccl_all_gather_through_the_c_abi_one_rank(amd, ctx):
    """safe_comm_* / safe_allgather_cols on a one-rank communicator (the 2-rank form runs in tests/test_gpu_multirank.py
    when two devices are visible): the slab arrives in the gathered buffer, on the context's stream."""
    from safepy_amd import backend as be
    uid = be.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = be.Comm(ctx, 1, 0, uid)
    src, dst = ctx.alloc(1 << 20), ctx.alloc(1 << 20)
    data = np.random.default_rng(3).integers(0, 255, size=1 << 20, dtype=np.uint8)
    src.upload(data)
    dst.zero()
    comm.allgather(src.ptr, 1 << 20, dst.ptr)
    ctx.sync()
    assert np.array_equal(dst.download((1 << 20,), dtype=np.uint8), data)
    comm.close()
    src.free()
    dst.free()


# ------------------------------------------------------------------ module functions ----

def test_compute_neighborhood_score_vs_reference(amd, golden_enr):
    g = golden_enr
    a = g['A'].astype(np.int64)
    for key, mat, kind in (('score_sum_q', g['b_q'], 'sum'), ('score_z_q', g['b_q'], 'z-score'),
                           ('score_z_q32', g['b_q_f32'], 'z-score'),
                           ('score_sum_binF', np.asfortranarray(g['b_bin'].astype(np.float32)), 'sum')):
        before = mat.copy()
        got = amd.compute_neighborhood_score(a, mat, kind)
        assert got.dtype == np.float64
        # f64 sums in a different order than OpenBLAS: 1e-6 is the north-star tolerance,
        # the observed agreement is ~1e-13
        np.testing.assert_allclose(got, g[key], rtol=1e-9, atol=1e-12, equal_nan=True)
        assert np.array_equal(np.isnan(got), np.isnan(g[key]))
        np.testing.assert_array_equal(mat, before)           # no mutation


def test_run_permutations_counts_vs_reference(amd, golden_enr):
    g = golden_enr
    a = g['A'].astype(np.int64)
    cn, cp = amd.run_permutations((a, g['b_bin'], 'sum', 25, 31), verbose=False)
    assert np.array_equal(cn, g['runperm_bin_neg']) and np.array_equal(cp, g['runperm_bin_pos'])
    cn, cp = amd.run_permutations((a, g['b_q'], 'sum', 25, 29), verbose=False)
    assert np.array_equal(cn, g['runperm_q_neg']) and np.array_equal(cp, g['runperm_q_pos'])


# ------------------------------------------------------------------ compute_pvalues ----

HYP = (('hyp_f64', 'b_bin', None, 'attribute_file'), ('hyp_f32F', 'b_bin', np.float32, 'attribute_file'),
       ('hyp_net', 'b_bin', None, 'network'))


@pytest.mark.parametrize('tag,src,dtype,bg', HYP)
def test_compute_pvalues_hypergeometric_vs_reference(amd, golden_nbr, golden_enr, tag, src, dtype, bg):
    g = golden_enr
    b = g[src].copy()
    if dtype is not None:
        b = np.asfortranarray(b.astype(dtype))
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.neighborhoods = g['A'].astype(np.int64)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(background=bg)
    assert sf.ns is None and sf.pvalues_neg is None           # hypergeometric path leaves them alone
    np.testing.assert_allclose(sf.pvalues_pos, g[tag + '_pvalues_pos'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(sf.nes, g[tag + '_nes'], rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(sf.nes_binary, g[tag + '_nes_binary'])
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, g[tag + '_num_enriched'])
    if bg == 'network':
        assert not np.isnan(sf.node2attribute).any()          # mutated in place like safe.py:451


def test_hypergeometric_forced_on_non_binary_gives_nan(amd, golden_nbr, golden_enr):
    g = golden_enr
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    sf.neighborhoods = g['A'].astype(np.int64)
    sf.load_attributes(attribute_file=g['b_int'].copy())
    sf.compute_pvalues(how='hypergeometric')
    np.testing.assert_array_equal(sf.pvalues_pos, g['hyp_int_pvalues_pos'])      # all NaN (K > population)
    np.testing.assert_array_equal(sf.nes, g['hyp_int_nes'])


RND = (('rnd_bin_sum', 'b_bin', np.float32, 'sum', 'both', 'attribute_file'),
       ('rnd_q_sum', 'b_q', None, 'sum', 'both', 'attribute_file'),
       ('rnd_q32_sum_hi', 'b_q_f32', None, 'sum', 'highest', 'attribute_file'),
       ('rnd_q_sum_lo', 'b_q', None, 'sum', 'lowest', 'attribute_file'),
       ('rnd_q_z', 'b_q', None, 'z-score', 'both', 'attribute_file'),
       ('rnd_q32_z', 'b_q_f32', None, 'z-score', 'both', 'attribute_file'),
       ('rnd_q_net', 'b_q', None, 'sum', 'both', 'network'),
       ('rnd_int_sum', 'b_int', None, 'sum', 'both', 'attribute_file'))


@pytest.mark.parametrize('tag,src,dtype,score,sign,bg', RND)
def test_compute_pvalues_randomization_vs_reference(amd, golden_nbr, golden_enr, tag, src, dtype, score, sign, bg):
    g = golden_enr
    b = g[src].copy()
    if dtype is not None:
        b = np.asfortranarray(b.astype(dtype))
    nperm, seed = (int(v) for v in g[tag + '_meta'])
    sf = _safe(amd, _layout_graph(amd, golden_nbr), attribute_sign=sign, random_seed=seed)
    sf.neighborhoods = g['A'].astype(np.int64)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(how='randomization', neighborhood_score_type=score, background=bg,
                       num_permutations=nperm, verbose=False)
    for arr in (sf.ns, sf.pvalues_neg, sf.pvalues_pos, sf.nes, sf.nes_binary):
        assert arr.dtype == np.float64 and arr.shape == b.shape
    np.testing.assert_allclose(sf.ns, g[tag + '_ns'], rtol=1e-9, atol=1e-12, equal_nan=True)
    # empirical p-values are counts / P: any flipped comparison would show as >= 1/P
    np.testing.assert_array_equal(sf.pvalues_neg, g[tag + '_pvalues_neg'])
    np.testing.assert_array_equal(sf.pvalues_pos, g[tag + '_pvalues_pos'])
    np.testing.assert_array_equal(sf.nes, g[tag + '_nes'])
    np.testing.assert_array_equal(sf.nes_binary, g[tag + '_nes_binary'])
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, g[tag + '_num_enriched'])


def test_define_then_compute_uses_device_resident_membership(amd, golden_nbr, golden_enr):
    """The usual flow: define_neighborhoods() then compute_pvalues() without the host
    ever materialising self.neighborhoods in between."""
    g = golden_enr
    sf = _safe(amd, _layout_graph(amd, golden_nbr), random_seed=11)
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    assert sf._neighborhoods_host is None
    sf.load_attributes(attribute_file=g['b_q'].copy())
    sf.compute_pvalues(how='randomization', num_permutations=40, verbose=False)
    np.testing.assert_array_equal(sf.pvalues_pos, g['rnd_q_sum_pvalues_pos'])
    np.testing.assert_array_equal(sf.nes, g['rnd_q_sum_nes'])
    assert np.array_equal(sf.neighborhoods, g['A'].astype(np.int64))


def test_config_errors_match_reference_behaviour(amd, golden_nbr):
    sf = _safe(amd, _layout_graph(amd, golden_nbr))
    with pytest.raises(ValueError):
        sf.define_neighborhoods(node_distance_metric='manhattan')
    assert sf.node_distance_metric == 'shortpath_weighted_layout'      # default restored (safe.py:205-209)
    sf.neighborhoods = np.eye(golden_nbr['xy'].shape[0], dtype=np.int64)
    sf.load_attributes(attribute_file=np.zeros((golden_nbr['xy'].shape[0], 2)))
    with pytest.raises(ValueError):
        sf.compute_pvalues(how='randomization', num_permutations=5)
    assert sf.num_permutations == 1000


# ------------------------------------------------- mid-size seeded check vs the oracle ----

def test_midsize_randomization_vs_oracle(amd, ctx):
    rng = np.random.default_rng(77)
    n, m, nperm = 1203, 150, 12
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.07)
    b = (rng.uniform(size=(n, m)) < 0.03).astype(np.float32)
    b[rng.choice(n, 60, replace=False)] = np.nan
    b = np.asfortranarray(b)
    want = orc.compute_pvalues(a, b.copy(order='F'), enrichment_type='randomization', num_permutations=nperm,
                               random_seed=3)
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.random_seed = 3
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.07)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(how='randomization', num_permutations=nperm, verbose=False)
    for key in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        np.testing.assert_array_equal(getattr(sf, key), want[key])     # integer-valued data: exact
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values,
                                  want['num_neighborhoods_enriched'])


@pytest.mark.parametrize('counts', ['bits', 'mfma', 'mfma-nolds', 'mfma-fused', 'per-element'])
def test_midsize_hypergeometric_vs_oracle(amd, ctx, monkeypatch, counts):
    """The forms of the hypergeometric path: bit-sliced counts + (n, K, X) table lookup, matrix-core
    counts + table lookup (split: packed u16 counts, then the streaming k_hyp_emit with the table slab
    in LDS; fused: lookup in the count kernel's epilogue), and the per-element tail evaluation."""
    if counts == 'per-element':
        monkeypatch.setenv('SAFE_HIP_HYPER_TABLE', '0')
    elif counts == 'mfma-fused':
        monkeypatch.setenv('SAFE_HIP_COUNTS', 'mfma')
        monkeypatch.setenv('SAFE_HIP_HYP_SPLIT', '0')
    elif counts == 'mfma-nolds':                       # table slab too large for LDS: lookups gathered from memory
        monkeypatch.setenv('SAFE_HIP_COUNTS', 'mfma')
        monkeypatch.setenv('SAFE_HIP_EMIT_LDS_KB', '0')
    else:
        monkeypatch.setenv('SAFE_HIP_COUNTS', counts)
    rng = np.random.default_rng(78)
    n, m = 1500, 300
    xy = rng.uniform(size=(n, 2))
    b = (rng.uniform(size=(n, m)) < rng.uniform(0.002, 0.3, size=m)).astype(np.float64)
    b[rng.choice(n, 40, replace=False)] = np.nan
    a = orc.neighborhoods_euclidean(xy, 0.12)
    want = orc.compute_pvalues(a, b.copy())
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.12)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues()
    want_kernel = {'bits': 'k_counts_bits<hypergeom>', 'mfma': 'k_hyp_emit',
                   'mfma-nolds': 'k_hyp_emit', 'mfma-fused': 'k_permtest_mfma<counts>', 'per-element': 'k_hypergeom_tail'}[counts]
    assert ctx.last_kernel()[0] == want_kernel
    np.testing.assert_allclose(sf.pvalues_pos, want['pvalues_pos'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-6, atol=1e-9)
    mism = (sf.nes_binary != want['nes_binary']).sum()
    assert mism == 0
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, want['num_neighborhoods_enriched'])


@pytest.mark.parametrize('n,m', [(257, 5), (300, 193), (1000, 1), (300, 194), (257, 6), (1100, 400)])
def test_split_hypergeometric_edge_shapes(amd, ctx, monkeypatch, n, m):
    """Split matrix-core form on awkward shapes: one row past a row group, fewer columns than one lane group,
    one column past a column group; columns without any annotation (every count 0: the table is cut at x = 0),
    a column annotating every node (K = N, p = 1 everywhere), an all-NaN row, isolated nodes; even and odd column counts,
    odd task lengths and ragged tiles."""
    monkeypatch.setenv('SAFE_HIP_COUNTS', 'mfma')
    rng = np.random.default_rng(n + m)
    xy = rng.uniform(size=(n, 2))
    xy[:3] += 10.0                                           # three nodes far away: neighborhoods of one
    b = (rng.uniform(size=(n, m)) < 0.1).astype(np.float64)
    b[:, 0] = 0.0
    if m > 2:
        b[:, 1] = 1.0
        b[:, 2] = 0.0
        b[7, 2] = 1.0
    b[11, :] = np.nan
    a = orc.neighborhoods_euclidean(xy, 0.02)
    want = orc.compute_pvalues(a, b.copy())
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.02)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues()
    assert ctx.last_kernel()[0] == 'k_hyp_emit'
    np.testing.assert_allclose(sf.pvalues_pos, want['pvalues_pos'], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(sf.nes_binary, want['nes_binary'])
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, want['num_neighborhoods_enriched'])


@pytest.mark.parametrize('counts', ['bits', 'mfma'])
def test_hypergeometric_p_on_the_threshold(amd, ctx, monkeypatch, counts):
    """p-values that sit exactly ON the enrichment threshold: a singleton attribute (K = 1) gives
    p = n / N, which is 0.05 for n = 9 of N = 180 -- SciPy returns 0.05 exactly and the reference
    calls such a neighborhood NOT enriched (`>` at safe.py:470).  The table kernel carries the tail in
    double-double and divides by the sum over the support, so p comes out correctly rounded; the
    binarisation is decided on p (common.h nes_p_cut), not on the device's log10.  Table p-values
    are within a few ulp of the exact rational value (SciPy itself is up to ~25 ulp off it)."""
    from fractions import Fraction
    from math import comb
    monkeypatch.setenv('SAFE_HIP_COUNTS', counts)
    n = 360 if counts == 'mfma' else 180
    rng = np.random.default_rng(9)
    # a ring of nodes: node i's neighborhood = the 9 (18) nodes around it -> n_i / N = 0.05 exactly
    width = n // 20
    a = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        a[i, (i + np.arange(width) - width // 2) % n] = 1
    b = np.zeros((n, 12))
    b[rng.choice(n, 12, replace=False), np.arange(12)] = 1          # singletons: K = 1
    b[:, 10] = rng.uniform(size=n) < 0.1
    b[:, 11] = rng.uniform(size=n) < 0.3
    want = orc.compute_pvalues(a, b.copy())
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(rng.uniform(size=(n, 2)))
    sf.neighborhoods = a
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues()
    assert ctx.last_kernel()[0] == {'bits': 'k_counts_bits<hypergeom>', 'mfma': 'k_hyp_emit'}[counts]
    on_threshold = want['pvalues_pos'] == 0.05
    assert on_threshold.sum() >= 10 * width
    assert np.array_equal(sf.pvalues_pos[on_threshold], want['pvalues_pos'][on_threshold])
    assert not sf.nes_binary[on_threshold].any() and not want['nes_binary'][on_threshold].any()
    np.testing.assert_array_equal(sf.nes_binary, want['nes_binary'])
    np.testing.assert_allclose(sf.pvalues_pos, want['pvalues_pos'], rtol=1e-13, atol=0)
    # against the exact rational tail
    x = a @ b
    k_col = b.sum(axis=0).astype(int)
    worst = 0.0
    for i in range(0, n, 7):
        for j in range(12):
            hi = min(k_col[j], width)
            num = sum(comb(int(k_col[j]), t) * comb(n - int(k_col[j]), width - t) for t in range(int(x[i, j]), hi + 1))
            exact = float(Fraction(num, comb(n, width)))
            if exact > 0:
                worst = max(worst, abs(sf.pvalues_pos[i, j] - exact) / np.spacing(exact))
    assert worst <= 1.0, worst


def test_torch_tensors_share_the_hip_runtime(amd, ctx):
    """Device pointers of torch tensors are usable by the library (one HIP runtime per
    process, see safepy_amd/_lib.py) and torch sees the library's results."""
    torch = pytest.importorskip('torch')
    assert torch.cuda.is_available()
    rng = np.random.default_rng(9)
    n = 700
    xy = rng.uniform(size=(n, 2))
    t_xy = torch.from_numpy(xy).to('cuda')
    t_mask = torch.empty((n, n), dtype=torch.int64, device='cuda')
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.euclidean_dense(t_xy.data_ptr(), n, 0.1, t_mask.data_ptr(), None)
        torch.cuda.synchronize()
        got = t_mask.cpu().numpy()
    finally:
        ctx.set_stream(None)
    assert np.array_equal(got, (orc.euclidean_distances(xy) < 0.1).astype(np.int64))


# ------------------------------------------ both forms of the permutation kernel ----------

@pytest.mark.parametrize('path', ['gather', 'scatter', 'bits'])
def test_binary_permutation_test_both_kernel_forms(amd, golden_enr, monkeypatch, path):
    """Binary attributes can run through the general f64 gather kernel or the sparse integer
    scatter kernel; both must reproduce the reference counts exactly."""
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
    g = golden_enr
    a = g['A'].astype(np.int64)
    cn, cp = amd.run_permutations((a, g['b_bin'], 'sum', 25, 31), verbose=False)
    assert np.array_equal(cn, g['runperm_bin_neg']) and np.array_equal(cp, g['runperm_bin_pos'])
    ctx = amd.Context.default(0)
    assert ctx.last_kernel()[0].startswith('k_permtest_' + path)


@pytest.mark.parametrize('path', ['gather', 'scatter', 'bits'])
def test_binary_randomization_asymmetric_membership(amd, monkeypatch, path):
    """User-supplied, non-symmetric membership (the scatter form walks the transpose)."""
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
    rng = np.random.default_rng(21)
    n, m, nperm = 333, 70, 15
    a = (rng.uniform(size=(n, n)) < 0.04).astype(np.int64)
    a[5, :] = 0                                              # an empty neighborhood
    a[:, 9] = 0                                              # a node nobody contains
    b = (rng.uniform(size=(n, m)) < 0.06).astype(np.float64)
    b[:, 3] = 0                                              # attribute with no annotations
    b[:, 4] = np.nan                                         # all-NaN attribute
    b[rng.choice(n, 20, replace=False)] = np.nan
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=8,
                               attribute_sign='highest')
    sf = amd.SAFE(verbose=False)
    sf.attribute_sign = 'highest'
    sf.random_seed = 8
    sf.neighborhoods = a
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(how='randomization', num_permutations=nperm, verbose=False)
    for key in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        np.testing.assert_array_equal(getattr(sf, key), want[key])


@pytest.mark.parametrize('path', ['scatter', 'bits'])
def test_integer_forms_large_support_and_many_permutations(amd, monkeypatch, path):
    """Support larger than one 256-row round per workgroup, > 255 permutations (counter
    carries beyond the low levels), dense and sparse columns."""
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
    rng = np.random.default_rng(22)
    n, m, nperm = 900, 12, 300
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.08)
    b = (rng.uniform(size=(n, m)) < np.linspace(0.02, 0.6, m)).astype(np.float32)
    cn_want, cp_want = orc.run_permutations(a, b, 'sum', nperm, 5)
    cn, cp = amd.run_permutations((a, b, 'sum', nperm, 5), verbose=False)
    assert np.array_equal(cn, cn_want) and np.array_equal(cp, cp_want)


@pytest.mark.parametrize('kernel,nperm', [('blk', 40), ('blk', 300), ('blk', 1), ('blk', 2), ('blk', 3), ('blk', 41), ('blk', 257), ('blk-plain', 41),
                                          ('blk-unpipelined', 41), ('pre', 40), ('barrier', 40)])
def test_bit_sliced_kernels_every_level_class(amd, monkeypatch, kernel, nperm):
    """The three bit-sliced kernels (blocked member lists = default, pre-permuted lists, permutation row in LDS) on a
    membership whose SELL slices fall into every width class of the blocked kernel (<= 8, <= 56, <= 248, <= 504, > 504 members:
    4 / 6 / 8 / 9 / 10 levels of the vertical sums), with columns dense enough that the sums really reach the top
    levels, an empty neighborhood, a ragged last word group; 300 permutations carry the counters past their low levels.  Odd and
    tiny permutation counts walk the blocked kernel's id stream through its tails (a slice of an odd number of blocks takes two
    permutations per period); 'blk-plain' is the form without the stream (SAFE_HIP_BITS_DBG=256, also the five-waves build's),
    'blk-unpipelined' the stream with a block's gathers all at its start (bit 9)."""
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', 'bits')
    if kernel == 'blk-plain':
        monkeypatch.setenv('SAFE_HIP_BITS_DBG', '256')
    elif kernel == 'blk-unpipelined':
        monkeypatch.setenv('SAFE_HIP_BITS_DBG', '512')
    kernel = kernel.split('-')[0]
    if kernel == 'pre':
        monkeypatch.setenv('SAFE_HIP_BITS_KERNEL', 'pre')
    elif kernel == 'barrier':
        monkeypatch.setenv('SAFE_HIP_BITS_PRE', '0')
    rng = np.random.default_rng(77)
    n, m = 1400, 131
    sizes = np.r_[rng.integers(600, 1000, 10), rng.integers(260, 500, 70), rng.integers(100, 248, 150), rng.integers(9, 56, 640),
                  rng.integers(0, 9, 530)]
    rng.shuffle(sizes)
    a = np.zeros((n, n), dtype=np.int64)
    for i, k in enumerate(sizes):
        a[i, rng.choice(n, k, replace=False)] = 1
    assert (a.sum(axis=1) == 0).any() and a.sum(axis=1).max() >= 600
    b = (rng.uniform(size=(n, m)) < np.linspace(0.005, 0.95, m)).astype(np.float32)
    b[rng.choice(n, 40, replace=False)] = np.nan
    b = np.asfortranarray(b)
    cn_want, cp_want = orc.run_permutations(a, b, 'sum', nperm, 13)
    cn, cp = amd.run_permutations((a, b, 'sum', nperm, 13), verbose=False)
    name = amd.Context.default(0).last_kernel()[0]
    assert name == {'blk': 'k_permtest_bits_blk', 'pre': 'k_permtest_bits_pre', 'barrier': 'k_permtest_bits'}[kernel]
    assert np.array_equal(cn, cn_want) and np.array_equal(cp, cp_want)
    ns = amd.compute_neighborhood_score(a, b, 'sum')
    assert np.array_equal(ns, orc.compute_neighborhood_score(a, b, 'sum'))


@pytest.mark.parametrize('biggest,expect', [(1023, 'k_permtest_bits_blk'), (1024, 'k_permtest_bits_pre'), (2047, 'k_permtest_bits_pre'),
                                            (2048, None)])
def test_hub_neighborhoods_of_1024_to_2047_members_stay_bit_sliced(amd, monkeypatch, biggest, expect):
    """A few hub neighborhoods decide the kernel family of a whole call: below 1024 members the blocked kernel (ten levels of the
    vertical sums), 1024 .. 2047 the sixteen-wave pre-permuted form with eleven levels (round 6), from 2048 the scatter / f64 /
    matrix-core kernels.  Counts against the oracle in every case; sums that reach the top level (dense columns)."""
    rng = np.random.default_rng(biggest)
    n, m, nperm = 2600, 70, 40
    sizes = np.r_[biggest, biggest - 1, rng.integers(900, biggest, 6), rng.integers(0, 300, n - 8)]
    rng.shuffle(sizes)
    a = np.zeros((n, n), dtype=np.int64)
    for i, k in enumerate(sizes):
        a[i, rng.choice(n, int(k), replace=False)] = 1
    b = (rng.uniform(size=(n, m)) < np.linspace(0.005, 0.99, m)).astype(np.float32)
    b[:, 3] = 1                                                  # every member carries it: the sums are the member counts
    b[rng.choice(n, 30, replace=False)] = np.nan
    cn_want, cp_want = orc.run_permutations(a, b, 'sum', nperm, 21)
    cn, cp = amd.run_permutations((a, b, 'sum', nperm, 21), verbose=False)
    name = amd.Context.default(0).last_kernel()[0]
    if expect is None:
        assert not name.startswith('k_permtest_bits'), name
    else:
        assert name == expect
    assert np.array_equal(cn, cn_want) and np.array_equal(cp, cp_want)
    assert np.array_equal(amd.compute_neighborhood_score(a, b, 'sum'), orc.compute_neighborhood_score(a, b, 'sum'))


# ------------------------------------------------------------ attribute sharding ----------

def test_sharded_columns_equal_unsharded(amd, ctx, golden_enr):
    """Column blocks computed separately (as ranks would) with the GLOBAL row flags reproduce the
    unsharded result; with per-shard flags they would not (a row whose only value lives in
    another shard must still move)."""
    from safepy_amd import backend as be, sharding
    g = golden_enr
    a = g['A'].astype(np.int64)
    b = g['b_q'].copy()
    n, m = b.shape
    b[11, :] = np.nan
    b[11, m - 1] = 0.75                                       # row 11 has a value only in the last shard
    nperm, seed = 20, 6
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=seed)
    nbr = be.Neighborhoods.from_dense(ctx, a)
    flags_global = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    got = {k: np.empty((n, m)) for k in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary')}
    for c0, c1 in sharding.column_shards(m, 3):
        attr = be.Attributes.from_host(ctx, np.ascontiguousarray(b[:, c0:c1]))
        attr.set_row_flags(flags_global)
        perms = be.Permutations(ctx, n, flags_global, nperm, seed)
        bufs = [ctx.alloc_f64(n, c1 - c0) for _ in range(5)] + [ctx.alloc_f64(c1 - c0)]
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [x.ptr for x in bufs])
        for k, buf in zip(('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'), bufs):
            got[k][:, c0:c1] = buf.download((n, c1 - c0))
        perms.close()
        attr.close()
    np.testing.assert_allclose(got['ns'], want['ns'], rtol=1e-9, atol=1e-12, equal_nan=True)
    for k in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        np.testing.assert_array_equal(got[k], want[k])


def test_column_range_arguments_of_the_c_abi(amd, ctx, golden_enr, monkeypatch):
    """col0/col1 select a block of a resident matrix (all three kernel forms)."""
    from safepy_amd import backend as be
    g = golden_enr
    a = g['A'].astype(np.int64)
    nbr = be.Neighborhoods.from_dense(ctx, a)
    for path, b in (('gather', g['b_q']), ('scatter', g['b_bin']), ('bits', g['b_bin'])):
        monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
        n, m = b.shape
        cn_w, cp_w = orc.run_permutations(a, b, 'sum', 15, 2)
        attr = be.Attributes.from_host(ctx, b)
        perms = be.Permutations(ctx, n, attr.row_flags(), 15, 2)
        c0, c1 = 5, 18
        neg, pos = ctx.alloc_f64(n, c1 - c0), ctx.alloc_f64(n, c1 - c0)
        be.permtest_counts(ctx, nbr, attr, perms, 'sum', None, neg.ptr, pos.ptr, col0=c0, col1=c1)
        assert np.array_equal(neg.download((n, c1 - c0)), cn_w[:, c0:c1]), path
        assert np.array_equal(pos.download((n, c1 - c0)), cp_w[:, c0:c1]), path
        perms.close()
        attr.close()


@pytest.mark.parametrize('score', ['sum', 'z-score'])
def test_f64_kernel_forms_agree(amd, ctx, golden_enr, monkeypatch, score):
    """Quantitative attributes: the LDS-resident f64 kernel (networks below one MFMA row group,
    z-scores) and the global-tile gather kernel add the members in the same order, so even the observed scores
    are bitwise identical."""
    from safepy_amd import backend as be
    g = golden_enr
    a = g['A'].astype(np.int64)
    b = g['b_q_f32']
    n, m = b.shape
    nbr = be.Neighborhoods.from_dense(ctx, a)
    res = {}
    for path in ('lds', 'gather'):
        monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
        attr = be.Attributes.from_host(ctx, b)
        perms = be.Permutations(ctx, n, attr.row_flags(), 33, 4)
        ns, neg, pos = ctx.alloc_f64(n, m), ctx.alloc_f64(n, m), ctx.alloc_f64(n, m)
        be.permtest_counts(ctx, nbr, attr, perms, score, ns.ptr, neg.ptr, pos.ptr)
        assert ctx.last_kernel()[0].startswith('k_permtest_' + path)
        res[path] = (ns.download((n, m)), neg.download((n, m)), pos.download((n, m)))
        perms.close()
        attr.close()
    for x, y in zip(res['lds'], res['gather']):
        np.testing.assert_array_equal(x, y)


# -------------------------------------------- integer exchange of the sharded result ----

@pytest.mark.parametrize('kind', ['binary', 'quantitative'])
def test_nes_from_gathered_packed_counts(amd, ctx, kind, monkeypatch):
    """What sharding.gather_nes does after the all-gather, with the gather simulated on one GPU:
    two column blocks computed separately, their packed counters laid end to end (one padded to
    the widest block), NES derived from the concatenation == NES of the unsharded run."""
    monkeypatch.setenv('SAFE_HIP_NARROW_LDS', '0')      # (blocks this narrow normally run on the f64 kernel, which leaves no integer counters)
    import torch
    from safepy_amd import backend as be, sharding
    rng = np.random.default_rng(3)
    n, m, nperm, seed = 700, 75, 40, 12
    xy = rng.uniform(size=(n, 2))
    b = (rng.uniform(size=(n, m)) < 0.05).astype(np.float64) if kind == 'binary' else rng.normal(size=(n, m))
    b[rng.choice(n, 30, replace=False)] = np.nan
    a = orc.neighborhoods_euclidean(xy, 0.1)
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    flags = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    shards = sharding.column_shards(m, 2)
    widest = max(c1 - c0 for c0, c1 in shards)
    slabs, layouts = [], []
    for c0, c1 in shards:
        attr = be.Attributes.from_host(ctx, np.ascontiguousarray(b[:, c0:c1]))
        attr.set_row_flags(flags)
        perms = be.Permutations(ctx, n, flags, nperm, seed)
        outs = [ctx.alloc_f64(n, c1 - c0) for _ in range(5)] + [ctx.alloc_f64(c1 - c0)]
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs])
        n_pad, m_loc, layout = be.packed_counts_info(ctx)
        assert layout in (0, 1) and m_loc == c1 - c0
        slab = torch.zeros(widest * n_pad, dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()                       # torch's fill and the context's copy run on different streams
        be.export_packed_counts(ctx, slab.data_ptr(), m_loc * n_pad)
        ctx.sync()
        slabs.append(slab)
        layouts.append((layout, n_pad))
        np.testing.assert_array_equal(outs[3].download((n, c1 - c0)), want['nes'][:, c0:c1])
        perms.close()
        attr.close()
    assert layouts[0] == layouts[1]
    everyone = torch.cat(slabs)
    full = torch.empty((n, 2 * widest), dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    be.nes_from_packed_counts(ctx, nbr, everyone.data_ptr(), layouts[0][0], layouts[0][1], 2 * widest, nperm, 'both',
                              full.data_ptr())
    full = full.cpu().numpy()
    got = np.concatenate([full[:, r * widest:r * widest + (c1 - c0)] for r, (c0, c1) in enumerate(shards)], axis=1)
    np.testing.assert_array_equal(got, want['nes'])
    nbr.close()


@pytest.mark.parametrize('kind', ['binary', 'quantitative'])
def test_every_output_from_gathered_packed_counts(amd, ctx, kind, monkeypatch):
    """safe_outputs_from_packed_counts: pvalues_neg, pvalues_pos, nes and nes_binary [N, M] from the concatenated integer
    counters of two column blocks == the unsharded call's matrices (safe.py:532-554, 468-472), any subset of them."""
    monkeypatch.setenv('SAFE_HIP_NARROW_LDS', '0')      # (blocks this narrow normally run on the f64 kernel, which leaves no integer counters)
    import torch
    from safepy_amd import backend as be, sharding
    rng = np.random.default_rng(5)
    n, m, nperm, seed = 650, 71, 40, 4
    xy = rng.uniform(size=(n, 2))
    b = (rng.uniform(size=(n, m)) < 0.06).astype(np.float64) if kind == 'binary' else rng.normal(size=(n, m))
    b[rng.choice(n, 25, replace=False)] = np.nan
    a = orc.neighborhoods_euclidean(xy, 0.1)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    flags = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    shards = sharding.column_shards(m, 2)                 # 36 + 35 columns: the compacted (ragged) form
    for sign in ('both', 'lowest'):
        want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=seed,
                                   attribute_sign=sign, enrichment_threshold=0.1)
        slabs = []
        for c0, c1 in shards:
            attr = be.Attributes.from_host(ctx, np.ascontiguousarray(b[:, c0:c1]))
            attr.set_row_flags(flags)
            perms = be.Permutations(ctx, n, flags, nperm, seed)
            outs = [ctx.alloc_f64(n, c1 - c0) for _ in range(5)] + [ctx.alloc_f64(c1 - c0)]
            be.randomization(ctx, nbr, attr, perms, 'sum', sign, 0.1, [o.ptr for o in outs])
            n_pad, m_loc, layout = be.packed_counts_info(ctx)
            slab = torch.zeros(m_loc * n_pad, dtype=torch.int32, device='cuda')
            torch.cuda.synchronize()
            be.export_packed_counts(ctx, slab.data_ptr(), m_loc * n_pad)
            ctx.sync()
            slabs.append(slab)
            perms.close()
            attr.close()
        everyone = torch.cat(slabs)
        names = ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary')
        for subset in (names, ('nes_binary',), ('pvalues_pos', 'nes')):
            full = {k: torch.full((n, m), -7.0, dtype=torch.float64, device='cuda') for k in subset}
            torch.cuda.synchronize()
            be.outputs_from_packed_counts(ctx, nbr, everyone.data_ptr(), layout, n_pad, m, nperm, sign, 0.1,
                                          [full[k].data_ptr() if k in full else None for k in names])
            for k in subset:
                np.testing.assert_array_equal(full[k].cpu().numpy(), want[k], err_msg='%s %s %s' % (kind, sign, k))
    with pytest.raises(amd.SafeHipError):
        be.outputs_from_packed_counts(ctx, nbr, everyone.data_ptr(), layout, n_pad, m, nperm, 'both', 0.1, [None] * 4)
    nbr.close()


def test_counts_outside_the_permutation_count_are_refused(amd, ctx):
    """safe_outputs_from_counts: counts are whole numbers in [0, num_permutations]; sums of ranks that each ran the full
    count, negative or NaN counts are an error (SAFE_E_VALUE), not an out-of-bounds table read."""
    from safepy_amd import backend as be
    n, m, nperm = 40, 9, 20
    rng = np.random.default_rng(0)
    ns = rng.normal(size=(n, m))
    ns[3, 4] = np.nan
    neg = rng.integers(0, nperm + 1, size=(n, m)).astype(np.float64)
    pos = nperm - neg
    neg[3, 4] = np.nan                                     # no test where the observed score is NaN: its counts do not matter
    d = [ctx.alloc_f64(n, m) for _ in range(7)] + [ctx.alloc_f64(m)]

    def run(cn, cp):
        d[0].upload(cn)
        d[1].upload(cp)
        d[2].upload(ns)
        be.outputs_from_counts(ctx, n, m, nperm, 'both', 0.05, d[0].ptr, d[1].ptr, d[2].ptr, [x.ptr for x in d[3:]])
        ctx.sync()
        return d[3].download((n, m)), d[4].download((n, m))

    p_neg, p_pos = run(neg, pos)
    ok = ~np.isnan(ns)
    assert np.array_equal(p_neg[ok], neg[ok] / nperm) and np.array_equal(p_pos[ok], pos[ok] / nperm) and np.isnan(p_neg[3, 4])
    for bad_value in (nperm + 1.0, 2.0 * nperm, -1.0, np.nan):
        bad = neg.copy()
        bad[7, 2] = bad_value
        with pytest.raises(amd.SafeHipError) as err:
            run(bad, pos)
        assert err.value.code == amd._lib.E_VALUE and 'outside' in str(err.value)
    run(neg, pos)                                          # the context is still usable


def test_more_than_65535_caller_supplied_permutations(amd, ctx):
    """A from-table / sliced handle with more rows than one grid dimension holds (the reference places no cap on
    num_permutations): 16-bit table copies are made in row blocks."""
    from safepy_amd import backend as be
    n, count = 12, 70000
    rng = np.random.default_rng(1)
    tables = np.argsort(rng.uniform(size=(count, n)), axis=1).astype(np.int32)
    perms = be.Permutations.from_table(ctx, tables)
    assert np.array_equal(perms.read(65530, 65545), tables[65530:65545])
    part = perms.slice(100, 69000)
    assert part.count == 68900 and np.array_equal(part.read(68000, 68900), tables[68100:69000])
    # the permutation test reads the 16-bit copy: counts over the LAST rows of the table against NumPy
    a = (rng.uniform(size=(n, n)) < 0.4).astype(np.int64)
    np.fill_diagonal(a, 1)
    b = (rng.uniform(size=(n, 3)) < 0.5).astype(np.float64)
    nbr = amd.Neighborhoods.from_dense(ctx, a)
    attr = be.Attributes.from_host(ctx, b)
    tail = perms.slice(69000, 70000)
    neg, pos = ctx.alloc_f64(n, 3), ctx.alloc_f64(n, 3)
    be.permtest_counts(ctx, nbr, attr, tail, 'sum', None, neg.ptr, pos.ptr)
    obs = a @ b
    want_n = sum(((a @ b[row]) <= obs) for row in tables[69000:70000]).astype(np.float64)
    assert np.array_equal(neg.download((n, 3)), want_n)
    for h in (tail, part, perms, attr, nbr):
        h.close()


def test_gather_nes_over_rccl_single_rank(amd, ctx):
    """The collective path itself (RCCL process group of one rank): packed counters are
    exported, all-gathered and turned into the NES matrix."""
    import torch
    import torch.distributed as dist
    from safepy_amd import backend as be, sharding
    created = False
    if not dist.is_initialized():
        import socket
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                device_id=torch.device('cuda', 0))
        created = True
    try:
        rng = np.random.default_rng(9)
        n, m, nperm, seed = 500, 40, 30, 2
        xy = rng.uniform(size=(n, 2))
        b = (rng.uniform(size=(n, m)) < 0.08).astype(np.float32)
        nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
        attr = be.Attributes.from_host(ctx, b)
        flags, stats = sharding.reduce_flags_and_stats(attr.row_flags(), attr.stats())
        assert stats['n_other'] == 0 and flags.sum() == n
        perms = be.Permutations(ctx, n, flags, nperm, seed)
        outs = [torch.empty((n, m), dtype=torch.float64, device='cuda') for _ in range(5)]
        enr = torch.empty(m, dtype=torch.float64, device='cuda')
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [t.data_ptr() for t in outs] + [enr.data_ptr()])
        full = sharding.gather_nes(ctx, nbr, outs[3], m, nperm, 'both')
        torch.cuda.synchronize()
        ctx.sync()
        assert torch.equal(full, outs[3])
        perms.close()
        attr.close()
        nbr.close()
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------- multiple_testing=True ----

@pytest.mark.parametrize('sort', ['auto', 'block', 'cub'])
@pytest.mark.parametrize('sign', ['both', 'highest'])
def test_fdr_randomization_vs_oracle(amd, sign, sort, monkeypatch):
    """safe.py:536-554 with multiple_testing=True: Benjamini-Hochberg per row, then NES and the
    binarised map from the adjusted p-values.  Empirical p-values are multiples of 1/P (many
    ties, zeros, ones), quantitative attributes, a NaN column under z-score.  Both row sorts: the
    bitonic network in LDS (rows up to 8192 attributes) and the library's segmented radix sort."""
    if sort != 'auto':                                           # 'auto': the sort-free histogram form where p = counts / P
        monkeypatch.setenv('SAFE_HIP_FDR_SORT', sort)
    rng = np.random.default_rng(61)
    n, m, nperm = 400, 150, 50
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.12)
    b = rng.normal(size=(n, m))
    b[rng.choice(n, 25, replace=False)] = np.nan
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=6,
                               attribute_sign=sign, multiple_testing=True)
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.random_seed = 6
    sf.attribute_sign = sign
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.12)
    sf.load_attributes(attribute_file=b.copy())
    sf.compute_pvalues(num_permutations=nperm, multiple_testing=True, verbose=False)
    np.testing.assert_array_equal(sf.pvalues_neg, want['pvalues_neg'])     # same divisions in the same order: bit-exact
    np.testing.assert_array_equal(sf.pvalues_pos, want['pvalues_pos'])
    np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-12, atol=1e-12)
    assert (sf.nes_binary != want['nes_binary']).sum() == 0
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, want['num_neighborhoods_enriched'])
    assert (sf.pvalues_pos < 1).any() and (sf.pvalues_pos == 1).any()


@pytest.mark.parametrize('sort', ['block', 'cub'])
def test_fdr_hypergeometric_vs_oracle_with_nan_rows(amd, sort, monkeypatch):
    """safe.py:599-608 with multiple_testing=True; a non-integer column makes hypergeom.sf NaN,
    and NumPy's minimum.accumulate then turns every adjusted p-value of those rows into NaN."""
    if sort != 'auto':                                           # 'auto': the sort-free histogram form where p = counts / P
        monkeypatch.setenv('SAFE_HIP_FDR_SORT', sort)
    rng = np.random.default_rng(62)
    n, m = 300, 90
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.15)
    b = (rng.uniform(size=(n, m)) < rng.uniform(0.01, 0.3, size=m)).astype(np.float64)
    b[rng.choice(n, 12, replace=False)] = np.nan
    for poison in (False, True):
        bb = b.copy()
        if poison:
            bb[5, 7] = 0.5                                       # K_j not an integer -> NaN p-values in column 7
        want = orc.compute_pvalues(a, bb.copy(), enrichment_type='hypergeometric', multiple_testing=True)
        sf = amd.SAFE(verbose=False)
        sf.graph = amd.LayoutGraph(xy)
        sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.15)
        sf.load_attributes(attribute_file=bb.copy())
        sf.compute_pvalues(how='hypergeometric', multiple_testing=True)
        assert np.array_equal(np.isnan(sf.pvalues_pos), np.isnan(want['pvalues_pos']))
        if poison:
            assert np.isnan(sf.pvalues_pos).all()
        np.testing.assert_allclose(sf.pvalues_pos, want['pvalues_pos'], rtol=1e-6, atol=1e-300)
        np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-6, atol=1e-9)
        assert (sf.nes_binary != want['nes_binary']).sum() == 0


@pytest.mark.parametrize('m', [1, 2, 3, 129, 4373])
def test_fdr_row_lengths_lds_equals_library_sort(amd, ctx, monkeypatch, m):
    """safe_fdr_adjust on random p-value rows with ties, zeros, ones and a NaN row: the in-LDS row kernel equals the
    library-sort path and the oracle bit for bit at awkward row lengths (1, 2, a power of two + 1, configs[1]'s 4373)."""
    import torch
    from safepy_amd import backend as be
    rng = np.random.default_rng(1000 + m)
    n = 37
    p = rng.integers(0, 41, size=(n, m)) / 40.0
    p[3, :] = rng.uniform(size=m)
    if m > 2:
        p[5, m // 2] = np.nan
    want = np.apply_along_axis(orc.fdrcorrection, 1, p)
    got = {}
    for sort in ('block', 'cub'):
        if sort == 'cub':
            monkeypatch.setenv('SAFE_HIP_FDR_SORT', 'cub')
        t = [torch.from_numpy(p.copy()).to('cuda'), torch.empty((n, m), dtype=torch.float64, device='cuda'),
             torch.empty((n, m), dtype=torch.float64, device='cuda'), torch.empty((m,), dtype=torch.float64, device='cuda')]
        torch.cuda.synchronize()
        be.fdr_adjust(ctx, n, m, 0, 'both', 0.05, [None] + [x.data_ptr() for x in t])
        ctx.sync()
        got[sort] = t[0].cpu().numpy()
    assert np.array_equal(got['block'], want, equal_nan=True)
    assert np.array_equal(got['cub'], want, equal_nan=True)


def test_numa_pinning_keeps_the_permutation_stream(amd, ctx):
    """Launcher helper: pin_threads_to_device_numa keeps the process on CPUs it was allowed before (or changes
    nothing), and the permutation stream of a handle made afterwards (its draw thread inherits the mask) is
    still NumPy's."""
    import os
    from safepy_amd import backend as be
    before = os.sched_getaffinity(0)
    node = be.pin_threads_to_device_numa(0)
    after = os.sched_getaffinity(0)
    assert after <= before and len(after) >= 1
    assert node is None or (isinstance(node, int) and node >= 0 and after != set())
    n, nperm, seed = 777, 300, 12345
    flags = np.ones(n, dtype=np.uint8)
    flags[::7] = 0
    perms = amd.Permutations(ctx, n, flags, nperm, seed)
    got = perms.read()
    perms.close()
    movable = np.flatnonzero(flags)
    np.random.seed(seed)
    cur = np.arange(n)
    for k in range(nperm):
        p = np.random.permutation(movable)                  # safe_extras.py:58
        cur[movable] = cur[p]                               # safe_extras.py:59-60 (cumulative)
        assert np.array_equal(got[k], cur), k
    os.sched_setaffinity(0, before)



@pytest.mark.parametrize('kind', ['binary-sum', 'quantitative-sum', 'quantitative-z'])
def test_more_permutations_than_the_lds_table_holds(amd, kind):
    """k_counts_finalize keeps the NES table in LDS up to 2559 permutations and reads it from memory beyond: the second form,
    for the integer counters (bit-sliced kernel) and for the direct ones of the f64 kernels (safe.py:528-554)."""
    rng = np.random.default_rng(77)
    n, m, nperm = 180, 70, 2700
    a = (rng.uniform(size=(n, n)) < 0.08).astype(np.int64)
    a |= a.T
    np.fill_diagonal(a, 1)
    if kind == 'binary-sum':
        b = (rng.uniform(size=(n, m)) < 0.1).astype(np.float64)
    else:
        b = np.round(rng.normal(size=(n, m)) * 8.0) / 8.0        # dyadic: sums and their comparisons are exact in any order
        b[:, 5] = np.nan
    b[rng.choice(n, 9, replace=False)] = np.nan
    score = 'z-score' if kind.endswith('-z') else 'sum'
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', neighborhood_score_type=score, num_permutations=nperm,
                               random_seed=3)
    sf = amd.SAFE(verbose=False)
    sf.random_seed = 3
    sf.neighborhoods = a
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(how='randomization', neighborhood_score_type=score, num_permutations=nperm, verbose=False)
    for key in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        np.testing.assert_array_equal(getattr(sf, key), want[key], err_msg=key)
    np.testing.assert_allclose(sf.ns, want['ns'], rtol=1e-12, atol=0, equal_nan=True)


def test_kernel_busy_time_is_the_union_of_the_launch_intervals(amd):
    """safe_last_kernel_busy_ms (bench.py's roofline.kernel_busy_ms_per_step): positive, and no more than the sum of the launches'
    durations that safe_last_kernel_stats reports (consecutive launches overlap on two streams)."""
    from safepy_amd import backend as be
    rng = np.random.default_rng(5)
    n, m, nperm = 600, 130, 300
    xy = rng.uniform(size=(n, 2))
    ctx = be.Context.default(0)
    nbr = be.Neighborhoods.euclidean(ctx, xy, 0.12)
    b = (rng.uniform(size=(n, m)) < 0.1).astype(np.float64)
    attr = be.Attributes.from_host(ctx, b)
    perms = be.Permutations(ctx, n, attr.row_flags(), nperm, 1)
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs])
    ctx.sync()
    name, avg_ms, launches = ctx.last_kernel()
    busy = ctx.last_kernel_busy_ms()
    perms.close()
    attr.close()
    assert name.startswith('k_permtest') and launches >= 1
    assert 0.0 < busy <= avg_ms * launches * 1.001


@pytest.mark.parametrize('nperm,m', [(10, 1), (10, 7), (40, 129), (1000, 4373), (5000, 300)])
def test_fdr_without_a_sort_equals_fdrcorrection(amd, nperm, m):
    """Randomization form of safe_fdr_adjust: p-values are counts / P, adjusted from the row's histogram over the counts instead
    of a sort (safe.py:536-542).  Against the oracle's fdrcorrection (itself equal to statsmodels' bit for bit): heavy ties, zeros,
    rows of ones, a single distinct value, and a row with a NaN (np.minimum.accumulate poisons the whole row)."""
    import torch
    from safepy_amd import backend as be
    ctx = amd.Context.default(0)
    rng = np.random.default_rng(nperm + m)
    n = 24
    counts = rng.integers(0, nperm + 1, size=(2, n, m))
    counts[0, 1] = nperm                                          # all ones
    counts[0, 2] = 0                                              # all zeros
    counts[0, 3] = rng.integers(0, 3, size=m)                     # very heavy ties
    counts[1, 4] = counts[1, 4, 0]                                # one distinct value
    p = counts.astype(np.float64) / float(nperm)
    p[0, 5, m // 2] = np.nan
    want = [np.stack([orc.fdrcorrection(row) for row in mat]) for mat in p]
    t = [torch.from_numpy(p[0].copy()).to('cuda'), torch.from_numpy(p[1].copy()).to('cuda'),
         torch.empty((n, m), dtype=torch.float64, device='cuda'), torch.empty((n, m), dtype=torch.float64, device='cuda'),
         torch.empty((m,), dtype=torch.float64, device='cuda')]
    torch.cuda.synchronize()
    be.fdr_adjust(ctx, n, m, nperm, 'both', 0.05, [x.data_ptr() for x in t])
    ctx.sync()
    for got, w in zip(t[:2], want):
        np.testing.assert_array_equal(got.cpu().numpy(), w)
    assert np.isnan(t[0].cpu().numpy()[5]).all()


def test_fdr_with_values_that_are_not_count_ratios(amd):
    """num_permutations > 0 with p-values that are NOT counts / num_permutations (a caller's own matrices): checked read-only
    first and adjusted through the sort -- nothing is refused and nothing is adjusted twice."""
    import torch
    from safepy_amd import backend as be
    ctx = amd.Context.default(0)
    n, m, nperm = 4, 33, 20
    p = np.random.default_rng(1).integers(0, nperm + 1, size=(n, m)).astype(np.float64) / nperm
    p[2, 7] = 0.123456
    q = np.random.default_rng(2).uniform(size=(n, m))
    t = [torch.from_numpy(p.copy()).to('cuda'), torch.from_numpy(q.copy()).to('cuda'),
         torch.empty((n, m), dtype=torch.float64, device='cuda'), torch.empty((n, m), dtype=torch.float64, device='cuda'),
         torch.empty((m,), dtype=torch.float64, device='cuda')]
    torch.cuda.synchronize()
    be.fdr_adjust(ctx, n, m, nperm, 'both', 0.05, [x.data_ptr() for x in t])
    np.testing.assert_array_equal(t[0].cpu().numpy(), orc.fdr_rows(p))
    np.testing.assert_array_equal(t[1].cpu().numpy(), orc.fdr_rows(q))
