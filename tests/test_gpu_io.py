"""Callers / data formats either side of the hot path on the device path:
`.scatter` networks + Euclidean pseudo-network, `calculate_edge_lengths`, `read_attributes`
(device alignment, census, resident handle) against the real reference's outputs
(tests/golden/io.npz) and against the oracle on larger seeded inputs.  Needs an MI355X."""
import gzip
import logging
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'io.npz')


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


@pytest.fixture(scope='module')
def g():
    return dict(np.load(GOLDEN))


def _write(tmp_path, name, data):
    p = os.path.join(str(tmp_path), name)
    with open(p, 'wb') as f:
        f.write(data.tobytes() if isinstance(data, np.ndarray) else data)
    return p


# ------------------------------------------------------------------ .scatter networks ----
def test_scatter_network_golden(amd, g, tmp_path):
    """load_network('.scatter') -> graph, nodes frame and pseudo-network equal the reference's;
    then the whole flow on it (euclidean neighborhoods, hypergeometric test, unimodality on the
    pseudo-network) equals the reference's outputs."""
    import pandas as pd
    path = _write(tmp_path, 'points.scatter', g['scatter_file'])
    sf = amd.SAFE(verbose=False)
    sf.neighborhood_radius = 0.07
    sf.load_network(network_file=path, node_key_attribute='key')
    assert np.array_equal([v for _, v in sf.graph.nodes.data('x')], g['scatter_x'])
    assert np.array_equal([v for _, v in sf.graph.nodes.data('y')], g['scatter_y'])
    assert list(sf.nodes['key']) == list(g['scatter_node_key'])
    assert list(sf.nodes['label']) == list(g['scatter_node_label'])
    assert np.array_equal(sf.nodes['id'].values, g['scatter_node_id'])
    e = np.array(sorted((min(u, v), max(u, v)) for u, v in sf.graph_euclidean.edges()), dtype=np.int64)
    assert np.array_equal(e, g['scatter_pseudo_edges'])
    assert {d['weight'] for _, _, d in sf.graph_euclidean.edges(data=True)} == {1.0}

    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.07)
    assert np.array_equal(sf.neighborhoods, g['scatter_neighborhoods'])
    n, m = g['scatter_attributes'].shape
    frame = pd.DataFrame(g['scatter_attributes'], index=['K%03d' % i for i in range(n)],
                         columns=['term %d' % j for j in range(m)])
    sf.load_attributes(attribute_file=frame)
    sf.compute_pvalues()
    assert np.array_equal(sf.nes_binary, g['scatter_nes_binary'])
    assert np.array_equal(sf.attributes['num_neighborhoods_enriched'].values, g['scatter_num_enriched'])
    sf.define_top_attributes()
    assert np.array_equal(sf.attributes['top'].values, g['scatter_top'].astype(bool))
    assert np.array_equal(sf.attributes['num_connected_components'].values, g['scatter_num_cc'])


def test_pseudo_network_arrays_form_and_oracle(amd):
    """Arrays-only pseudo-network on a larger layout == the oracle's dense pdist form."""
    from safepy_amd import safe_io
    rng = np.random.default_rng(4)
    xy = rng.normal(size=(1500, 2)) * np.array([2.0, 0.5])
    lg = safe_io.euclidean_pseudo_network(amd.LayoutGraph(xy), 0.03, as_networkx=False)
    got = np.stack([lg.edge_u, lg.edge_v], axis=1)
    want = orc.pseudo_network_edges(xy, 0.03)
    assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want)
    assert (lg.edge_u == lg.edge_v).sum() == 1500          # one self loop per node


# --------------------------------------------------------------- calculate_edge_lengths ----
def test_calculate_edge_lengths_weighted_golden(amd, g):
    import networkx as nx
    from safepy_amd import safe_io
    xy = g['wl_xy']
    G = nx.Graph()
    for i in range(xy.shape[0]):
        G.add_node(i, x=float(xy[i, 0]), y=float(xy[i, 1]))
    for u, v, w in zip(g['wl_edge_u'], g['wl_edge_v'], g['wl_weight']):
        G.add_edge(int(u), int(v), weight=float(w))
    G = safe_io.calculate_edge_lengths(G, verbose=False)
    got = np.array([G.edges[int(u), int(v)].get('length', np.nan) for u, v in zip(g['wl_edge_u'], g['wl_edge_v'])])
    assert np.array_equal(got, g['wl_length'], equal_nan=True)


def test_calculate_edge_lengths_unweighted_and_layoutgraph(amd, golden_nbr):
    import networkx as nx
    from safepy_amd import safe_io
    gn = golden_nbr
    xy, eu, ev = gn['xy'], gn['edge_u'], gn['edge_v']
    G = nx.Graph()
    for i in range(xy.shape[0]):
        G.add_node(i, x=float(xy[i, 0]), y=float(xy[i, 1]))
    G.add_edges_from(zip(eu.tolist(), ev.tolist()))
    safe_io.calculate_edge_lengths(G, verbose=False)
    assert np.array_equal([G.edges[int(u), int(v)]['length'] for u, v in zip(eu, ev)], gn['edge_length'])
    lg = safe_io.calculate_edge_lengths(amd.LayoutGraph(xy, eu, ev), verbose=False)
    assert np.array_equal(lg.length, gn['edge_length'])
    # relabelled nodes: the reference's matrix-index addressing does not apply -> refuse loudly
    H = nx.relabel_nodes(G, {0: 'a'})
    with pytest.raises(ValueError):
        safe_io.calculate_edge_lengths(H, verbose=False)


# ------------------------------------------------------------------------ read_attributes ----
@pytest.mark.parametrize('tag,ext', [('ra_bin', '.txt'), ('ra_q', '.txt.gz'), ('ra_f32', '.txt')])
def test_read_attributes_files_golden(amd, g, tmp_path, tag, ext):
    from safepy_amd import safe_io
    path = _write(tmp_path, tag + ext, g[tag + '_file'])
    attributes, order, mat = safe_io.read_attributes(attribute_file=path, node_label_order=list(g['ra_node_order']),
                                                     verbose=False)
    want = g[tag + '_matrix']
    assert mat.dtype == want.dtype and mat.shape == want.shape
    assert np.array_equal(mat, want, equal_nan=True)
    assert [mat.flags['F_CONTIGUOUS'], mat.flags['C_CONTIGUOUS']] == list(g[tag + '_forder'])
    assert list(attributes['name']) == list(g[tag + '_names'])
    assert np.array_equal(attributes['id'].values, g[tag + '_ids'])
    assert order == list(g['ra_node_order'])


def test_read_attributes_dataframe_golden(amd, g):
    import pandas as pd
    from safepy_amd import safe_io
    frame = pd.DataFrame(g['ra_df_values'], index=list(g['ra_df_index']), columns=list('abcdef'))
    np.random.seed(3)
    attributes, order, mat = safe_io.read_attributes(attribute_file=frame.copy(), node_label_order=list(g['ra_node_order']),
                                                     mask_duplicates=True, fill_value=0, verbose=False)
    assert np.array_equal(mat, g['ra_df_matrix'], equal_nan=True)
    assert [mat.flags['F_CONTIGUOUS'], mat.flags['C_CONTIGUOUS']] == list(g['ra_df_forder'])
    assert list(attributes['name']) == list(g['ra_df_names'])
    attributes, order, mat = safe_io.read_attributes(attribute_file=frame.copy(), verbose=False)
    assert np.array_equal(mat, g['ra_df_noorder_matrix'])
    assert list(order) == list(g['ra_df_noorder_order'])


@pytest.mark.parametrize('dtype,order', [(np.float32, 'C'), (np.float32, 'F'), (np.float64, 'C'), (np.float64, 'F')])
def test_reindex_kernel_vs_oracle(amd, dtype, order):
    """The alignment kernel at ragged sizes (not multiples of the 64 x 64 tile), both table orders,
    both output orders, absent (-1) and masked (-2) rows, against explicit NumPy indexing."""
    rng = np.random.default_rng(11)
    ctx = amd.Context.default(0)
    for (nl, m, n) in ((1, 1, 1), (70, 3, 129), (517, 203, 1001), (64, 64, 64)):
        table = np.asarray(rng.normal(size=(nl, m)).astype(dtype), order=order)
        table[rng.uniform(size=table.shape) < 0.05] = np.nan
        row_map = rng.integers(-2, nl, size=n)
        for out_order in ('C', 'F'):
            attr, host = amd.Attributes.reindexed(ctx, table, row_map, fill_value=-7.5, order=out_order)
            want = np.where((row_map >= 0)[:, None], table[np.maximum(row_map, 0)], dtype(-7.5))
            want[row_map == -2] = np.nan
            assert host.dtype == dtype and host.flags[out_order + '_CONTIGUOUS']
            assert np.array_equal(host, want, equal_nan=True)
            assert np.array_equal(attr.download(dtype, out_order), want, equal_nan=True)
            assert attr.value_counts() == orc.value_census(want.astype(np.float64))
            attr.nan_to_zero()
            assert np.array_equal(attr.download(dtype, out_order), np.nan_to_num(want, nan=0.0))
            assert attr.value_counts()[0] == 0
            attr.close()


def test_read_attributes_large_vs_oracle_and_census_log(amd, tmp_path, caplog):
    """A GO-sized text file (2000 labels x 300 terms, gzip) through the device path == the oracle's
    explicit alignment; the logged value census equals the oracle's counts."""
    import pandas as pd
    from safepy_amd import safe_io
    rng = np.random.default_rng(21)
    nl, m, n = 2000, 300, 2500
    labels = ['Y%05d' % i for i in rng.permutation(4000)[:nl]]
    b = (rng.uniform(size=(nl, m)) < 0.02).astype(int)
    body = '\n'.join(['ORF\t' + '\t'.join('GO:%d' % j for j in range(m))] +
                     [lab + '\t' + '\t'.join(map(str, row)) for lab, row in zip(labels, b)]) + '\n'
    path = os.path.join(str(tmp_path), 'go.txt.gz')
    with gzip.open(path, 'wb') as f:
        f.write(body.encode())
    node_order = ['Y%05d' % i for i in rng.integers(0, 4000, size=n)]
    with caplog.at_level(logging.INFO):
        attributes, order, mat = safe_io.read_attributes(attribute_file=path, node_label_order=list(node_order), verbose=True)
    names, table = orc.parse_attribute_text(path)
    _, want = orc.align_attributes(table, list(node_order))
    assert mat.dtype == want.dtype == np.float32
    assert np.array_equal(mat, want, equal_nan=True)
    census = orc.value_census(want.astype(np.float64))
    text = caplog.text
    for label, count in zip(('NaNs', 'zeros', 'positives', 'negatives'), census):
        assert 'Values: %d %s' % (count, label) in text


def test_resident_attributes_equal_uploaded(amd, g, tmp_path):
    """load_attributes(keep_on_device=True): compute_pvalues on the resident matrix == the default
    upload-per-call flow, for both backgrounds; the host mirror is read-only and follows the
    in-place NaN -> 0 of background='network' (safe.py:449-451)."""
    path = _write(tmp_path, 'ra_bin.txt', g['ra_bin_file'])
    rng = np.random.default_rng(2)
    n = len(g['ra_node_order'])
    xy = rng.uniform(size=(n, 2))

    def run(keep, background):
        sf = amd.SAFE(verbose=False)
        sf.graph = amd.LayoutGraph(xy, keys=list(g['ra_node_order']))
        sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.15)
        sf.load_attributes(attribute_file=path, keep_on_device=keep)
        sf.compute_pvalues(background=background)
        return sf

    for background in ('attribute_file', 'network'):
        a, b = run(False, background), run(True, background)
        assert b._resident_attributes() is not None and not b.node2attribute.flags.writeable
        assert a._resident_attributes() is None and a.node2attribute.flags.writeable
        assert np.array_equal(a.node2attribute, b.node2attribute, equal_nan=True)
        assert np.array_equal(a.pvalues_pos, b.pvalues_pos, equal_nan=True)
        assert np.array_equal(a.nes, b.nes, equal_nan=True)
        assert np.array_equal(a.nes_binary, b.nes_binary, equal_nan=True)
        if background == 'network':
            assert not np.isnan(b.node2attribute).any()
        else:
            assert np.isnan(b.node2attribute).any()
        # a second call reuses the resident matrix; replacing the host array drops it
        b.compute_pvalues(background=background)
        assert np.array_equal(a.nes, b.nes, equal_nan=True)
        b.node2attribute = np.array(b.node2attribute)
        assert b._resident_attributes() is None
        b.compute_pvalues(background=background)
        assert np.array_equal(a.nes, b.nes, equal_nan=True)


# ------------------------------------------------------------------ batch driver (CLI) ----
def _batch_inputs(g, tmp_path, quantitative):
    net = _write(tmp_path, 'points.scatter', g['scatter_file'])
    n = len(g['scatter_x'])
    rng = np.random.default_rng(8)
    m = 23
    keys = ['K%03d' % i for i in range(n)]
    if quantitative:
        rows = [[k] + ['%.4f' % v for v in rng.normal(size=m)] for k in keys[:170]]
    else:
        rows = [[k] + [str(int(v)) for v in (rng.uniform(size=m) < 0.2)] for k in keys[:170]]
    body = '\n'.join(['\t'.join(['ORF'] + ['t%d' % j for j in range(m)])] + ['\t'.join(r) for r in rows]) + '\n'
    attr = _write(tmp_path, 'attrs_q.txt' if quantitative else 'attrs_b.txt', body.encode())
    return net, attr


@pytest.mark.parametrize('quantitative', [False, True])
def test_run_batch_single_gpu_equals_safe(amd, g, tmp_path, quantitative):
    """python -m safepy_amd.run_batch on one GPU writes <attribute_file>_safe_nes.p (safe.py:1357)
    holding the NES matrix SAFE.compute_pvalues produces for the same inputs."""
    import pickle
    from safepy_amd import run_batch
    net, attr = _batch_inputs(g, tmp_path, quantitative)
    assert run_batch.main([attr, '--network', net, '--radius', '0.07', '--permutations', '50', '--seed', '5']) == 0
    with open(attr + '_safe_nes.p', 'rb') as f:
        got = pickle.load(f)
    sf = amd.SAFE(verbose=False)
    sf.random_seed = 5
    sf.neighborhood_radius = 0.07
    sf.load_network(network_file=net, node_key_attribute='key')
    sf.define_neighborhoods(node_distance_metric='euclidean')
    sf.load_attributes(attribute_file=attr)
    sf.compute_pvalues(num_permutations=50)
    assert (sf.pvalues_neg is None) == (not quantitative)      # binary -> hypergeometric, quantitative -> permutations
    assert np.array_equal(got, sf.nes, equal_nan=True)
    want = orc.compute_pvalues(sf.neighborhoods, sf.node2attribute.copy(), num_permutations=50, random_seed=5)
    np.testing.assert_allclose(got, want['nes'], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('quantitative', [False, True])
def test_sharded_compute_pvalues_one_rank_rccl(amd, g, tmp_path, quantitative):
    """sharding.sharded_compute_pvalues over a one-rank RCCL group == SAFE.compute_pvalues (both
    branches of the whole-matrix 'auto' rule, the collectives really run)."""
    import socket
    import torch
    import torch.distributed as dist
    from safepy_amd import sharding
    net, attr = _batch_inputs(g, tmp_path, quantitative)
    sf = amd.SAFE(verbose=False)
    sf.random_seed = 11
    sf.neighborhood_radius = 0.07
    sf.load_network(network_file=net, node_key_attribute='key', pseudo_network='arrays')
    sf.define_neighborhoods(node_distance_metric='euclidean')
    sf.load_attributes(attribute_file=attr)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        out = sharding.sharded_compute_pvalues(sf._ctx(), sf._device_neighborhoods(), np.ascontiguousarray(sf.node2attribute),
                                               sf.node2attribute.shape[1], num_permutations=40, random_seed=11,
                                               gather=('nes', 'nes_binary'))
    finally:
        dist.destroy_process_group()
    sf.compute_pvalues(num_permutations=40)
    assert out['how'] == ('randomization' if quantitative else 'hypergeometric')
    assert np.array_equal(out['full_nes'], sf.nes, equal_nan=True)
    assert np.array_equal(out['full_nes_binary'], sf.nes_binary, equal_nan=True)
    assert np.array_equal(out['num_neighborhoods_enriched'], sf.attributes['num_neighborhoods_enriched'].values)


@pytest.mark.parametrize('quantitative', [False, True])
def test_sharded_fdr_equals_unsplit(amd, g, tmp_path, quantitative):
    """multiple_testing=True through the sharded driver (one-rank RCCL group: gather, whole-matrix
    Benjamini-Hochberg, NES / nes_binary / enriched counts rebuilt) == SAFE.compute_pvalues(multiple_testing=True)."""
    import socket
    import torch
    import torch.distributed as dist
    from safepy_amd import sharding
    net, attr = _batch_inputs(g, tmp_path, quantitative)
    sf = amd.SAFE(verbose=False)
    sf.random_seed = 11
    sf.neighborhood_radius = 0.07
    sf.load_network(network_file=net, node_key_attribute='key', pseudo_network='arrays')
    sf.define_neighborhoods(node_distance_metric='euclidean')
    sf.load_attributes(attribute_file=attr)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        out = sharding.sharded_compute_pvalues(sf._ctx(), sf._device_neighborhoods(), np.ascontiguousarray(sf.node2attribute),
                                               sf.node2attribute.shape[1], num_permutations=40, random_seed=11,
                                               gather=('nes', 'nes_binary', 'pvalues_pos'), multiple_testing=True)
    finally:
        dist.destroy_process_group()
    sf.compute_pvalues(num_permutations=40, multiple_testing=True)
    assert np.array_equal(out['full_pvalues_pos'], sf.pvalues_pos, equal_nan=True)
    assert np.array_equal(out['pvalues_pos'], sf.pvalues_pos, equal_nan=True)          # one rank: the local block is everything
    assert np.array_equal(out['full_nes'], sf.nes, equal_nan=True)
    assert np.array_equal(out['full_nes_binary'], sf.nes_binary, equal_nan=True)
    assert np.array_equal(out['num_neighborhoods_enriched'], sf.attributes['num_neighborhoods_enriched'].values)
    if quantitative:
        assert np.array_equal(out['pvalues_neg'], sf.pvalues_neg, equal_nan=True)
    sf.compute_pvalues(num_permutations=40, multiple_testing=False)
    assert not np.array_equal(out['full_pvalues_pos'], sf.pvalues_pos, equal_nan=True)      # the adjustment did something


# ------------------------------------------------------------------ lazy result attributes ----
def test_lazy_outputs_equal_eager_and_pickle(amd, golden_nbr, golden_enr):
    """compute_pvalues() leaves the result matrices on the device; each is copied on its first read
    and behaves as the plain ndarray attribute of the reference from then on (same values as the
    eager copies, settable, picklable while still on the device)."""
    import pickle
    g = golden_enr
    xy = golden_nbr['xy']

    def run(lazy, **kw):
        sf = amd.SAFE(verbose=False)
        sf.lazy_outputs = lazy
        sf.random_seed = 11
        sf.graph = amd.LayoutGraph(xy)
        sf.neighborhoods = g['A'].astype(np.int64)
        sf.load_attributes(attribute_file=g['b_q'].copy())
        sf.compute_pvalues(num_permutations=40, **kw)
        return sf

    a, b = run(True), run(False)
    assert type(b.__dict__['_r_nes']) is np.ndarray and type(a.__dict__['_r_nes']).__name__ == '_DeviceResult'
    blob = pickle.dumps(a)                                   # materialises what is still on the device
    for name in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        x, y = getattr(a, name), getattr(b, name)
        assert type(x) is np.ndarray and x.dtype == np.float64 and x.flags['C_CONTIGUOUS']
        assert np.array_equal(x, y, equal_nan=True)
        assert getattr(a, name) is x                         # the copy happens once
        assert np.array_equal(getattr(pickle.loads(blob), name), y, equal_nan=True)
    a.nes = None
    assert a.nes is None
    # a second run replaces results that were never read (their device buffers are released)
    c = run(True)
    c.compute_pvalues(num_permutations=40)
    assert np.array_equal(c.nes, b.nes, equal_nan=True)
    # hypergeometric path: ns / pvalues_neg keep whatever they held (None here), like the reference
    h = amd.SAFE(verbose=False)
    h.graph = amd.LayoutGraph(xy)
    h.neighborhoods = g['A'].astype(np.int64)
    h.load_attributes(attribute_file=g['b_bin'].copy())
    h.compute_pvalues()
    assert h.ns is None and h.pvalues_neg is None
    np.testing.assert_array_equal(h.nes_binary, g['hyp_f64_nes_binary'])


# ------------------------------------------------------------------ device buffer pool ----
def test_buffer_pool_is_bounded_reuses_and_evicts_least_recently_used(amd, monkeypatch):
    """Released result buffers are kept for reuse up to a share of the device memory; beyond it the sizes
    unused the longest go back to the driver (a run over differently shaped attribute files must not pile up dead sizes)."""
    from safepy_amd import backend as be
    ctx = amd.Context.default(0)
    ctx.trim()
    assert ctx._pool_bytes == 0
    monkeypatch.setattr(be.DeviceBuffer, 'POOL_FRACTION', (8 << 20) / ctx.hbm_bytes)
    a, b = ctx.alloc(3 << 20), ctx.alloc(4 << 20)
    b_ptr = b.ptr
    a.free()
    b.free()
    assert ctx._pool_bytes == 7 << 20
    c = ctx.alloc(2 << 20)
    c.free()                                   # 9 MiB > 8 MiB: the 3 MiB buffer (released first) is evicted
    assert sorted(ctx._pool) == [2 << 20, 4 << 20] and ctx._pool_bytes == 6 << 20
    d = ctx.alloc(4 << 20)
    assert d.ptr == b_ptr and ctx._pool_bytes == 2 << 20      # same size: the pooled buffer comes back
    d.free()
    small = ctx.alloc(1 << 10)
    small.free()                               # below 1 MiB: never pooled
    assert ctx._pool_bytes == 6 << 20
    huge = ctx.alloc(9 << 20)
    huge.free()                                # larger than the whole pool: straight back to the driver
    assert ctx._pool_bytes == 6 << 20
    ctx.trim()
    assert ctx._pool_bytes == 0 and not ctx._pool
