#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference.

Run in the build container only (needs /root/reference, networkx, pandas):

    python tests/golden/make_golden.py

It imports baryshnikova-lab/safepy from /root/reference, pushes seeded synthetic inputs
through ``SAFE.define_neighborhoods`` / ``SAFE.compute_pvalues`` /
``safe_extras.run_permutations`` and stores inputs + outputs as small ``.npz``
files next to this script.  Only data is written: no reference source travels.

statsmodels (the reference imports ``fdrcorrection`` from it, safe.py:30) is not installed
for the system interpreter, but a pure-Python-on-this-path copy (0.12.2) sits in the image at
/opt/conda/lib/python3.9/site-packages: that directory is APPENDED to sys.path, so numpy /
scipy / pandas stay the system ones and only statsmodels (+ patsy) resolve there.  The
reference pins 0.14.4 (extras/requirements.txt); the version actually used is recorded in
``fdr.npz`` (``statsmodels_version``).  Only if that import fails is a raising stub installed
(then ``fdr`` cannot be generated).

    python tests/golden/make_golden.py          # everything
    python tests/golden/make_golden.py fdr      # only fdr.npz  (multiple_testing=True)
    python tests/golden/make_golden.py big      # only big.npz  (N = 1200, 300 permutations)
    python tests/golden/make_golden.py io       # only io.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


STATSMODELS_SITE = '/opt/conda/lib/python3.9/site-packages'


def import_reference():
    if STATSMODELS_SITE not in sys.path:
        sys.path.append(STATSMODELS_SITE)      # append: numpy / scipy / pandas stay the system ones
    try:
        import statsmodels.stats.multitest     # noqa: F401  the real one (0.12.2 in this image)
    except Exception:                          # no statsmodels anywhere: multiple_testing=False only
        sm = types.ModuleType('statsmodels')
        sms = types.ModuleType('statsmodels.stats')
        smm = types.ModuleType('statsmodels.stats.multitest')

        def fdrcorrection(*a, **k):
            raise RuntimeError('statsmodels is stubbed; multiple_testing=True cannot be generated')
        smm.fdrcorrection = fdrcorrection
        sm.stats = sms
        sms.multitest = smm
        sys.modules['statsmodels'] = sm
        sys.modules['statsmodels.stats'] = sms
        sys.modules['statsmodels.stats.multitest'] = smm
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'safepy'))
    import logging
    logging.disable(logging.CRITICAL)
    from safepy import safe, safe_extras, safe_io
    return safe, safe_extras, safe_io


def clustered_layout(rng, n, n_blobs=6, spread=0.07):
    """Mixture of Gaussian blobs + uniform background: ragged neighborhood sizes."""
    centers = rng.uniform(0.15, 0.85, size=(n_blobs, 2))
    which = rng.integers(0, n_blobs + 1, size=n)
    xy = rng.uniform(0.0, 1.0, size=(n, 2))
    in_blob = which < n_blobs
    xy[in_blob] = centers[which[in_blob]] + rng.normal(0.0, spread, size=(int(in_blob.sum()), 2))
    return xy


def radius_graph_edges(xy, r, rng, keep=0.6):
    d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    iu, ju = np.nonzero(np.triu(d < r, k=1))
    sel = rng.uniform(size=iu.size) < keep
    return iu[sel].astype(np.int64), ju[sel].astype(np.int64)


def make_graph(nx, xy, eu, ev):
    g = nx.Graph()
    for i in range(xy.shape[0]):
        g.add_node(i, x=float(xy[i, 0]), y=float(xy[i, 1]), label='n%d' % i, label_orf='ORF%d' % i)
    for u, v in zip(eu, ev):
        g.add_edge(int(u), int(v))
    return g


def new_safe(safe, graph, **attrs):
    sf = safe.SAFE(verbose=False)
    sf.graph = graph
    for k, v in attrs.items():
        setattr(sf, k, v)
    return sf


def set_attributes(pd, sf, mat):
    sf.node2attribute = mat
    sf.attributes = pd.DataFrame({'id': np.arange(mat.shape[1]),
                                  'name': ['attr%d' % j for j in range(mat.shape[1])]})


def make_domains(safe, safe_io, nx, pd):
    """define_top_attributes / define_domains / trim_domains of the real reference (safe.py:610-745)
    on a small network with spatially coherent binary attributes."""
    import warnings
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(5)
    n, m = 300, 40
    xy = clustered_layout(rng, n)
    eu, ev = radius_graph_edges(xy, 0.08, rng)
    g = make_graph(nx, xy, eu, ev)
    g = safe_io.calculate_edge_lengths(g, verbose=False)
    b = np.zeros((n, m))
    for j in range(m):
        c = xy[rng.integers(n)]
        d = np.sqrt(((xy - c) ** 2).sum(1))
        b[:, j] = (d < rng.uniform(0.05, 0.15)) & (rng.uniform(size=n) < 0.8)
    names = ['%s %s of the %s' % (rng.choice(['dna', 'rna', 'protein']), rng.choice(['repair', 'transport', 'folding', 'splicing']),
                                   rng.choice(['nucleus', 'membrane', 'cytosol'])) for _ in range(m)]
    out = {'xy': xy, 'edge_u': eu, 'edge_v': ev, 'attributes': b, 'names': np.array(names)}
    sf = new_safe(safe, g)
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    sf.node2attribute = b.copy()
    sf.attributes = pd.DataFrame({'id': np.arange(m), 'name': names})
    sf.compute_pvalues()
    out['nes'] = sf.nes
    out['nes_binary'] = sf.nes_binary
    out['num_enriched'] = sf.attributes['num_neighborhoods_enriched'].values.astype(np.float64)
    sf.define_top_attributes()
    a = sf.attributes
    out['top'] = a['top'].values.astype(np.int8)
    out['num_cc'] = a['num_connected_components'].values.astype(np.int64)
    out['num_large_cc'] = a['num_large_connected_components'].values.astype(np.int64)
    # pandas' .at stores a one-component array as a 0-d array; non-candidates hold None / NaN
    cc = [np.atleast_1d(s) if isinstance(s, np.ndarray) else None for s in a['size_connected_components']]
    width = max(len(s) for s in cc if s is not None)
    sizes = np.full((m, width), -1, dtype=np.int64)                # -1 padded; row of -1 = not a candidate
    for j, s in enumerate(cc):
        if s is not None:
            sizes[j, :len(s)] = s
    out['cc_sizes'] = sizes
    for thr in (0.75, 0.65):
        sf.define_domains(attribute_distance_threshold=thr)
        tag = 'thr%g_' % thr
        out[tag + 'domain'] = sf.attributes['domain'].values.astype(np.int64)
        dom_cols = [c for c in sf.node2domain.columns if c not in ('primary_domain', 'primary_nes')]
        out[tag + 'domain_ids'] = np.array(dom_cols, dtype=np.int64)
        out[tag + 'node2domain'] = sf.node2domain[dom_cols].values
        out[tag + 'primary_domain'] = sf.node2domain['primary_domain'].values.astype(np.int64)
        out[tag + 'primary_nes'] = sf.node2domain['primary_nes'].values
    sf.trim_domains()
    out['trim_domain'] = sf.attributes['domain'].values.astype(np.int64)
    out['trim_primary_domain'] = sf.node2domain['primary_domain'].values.astype(np.int64)
    out['trim_primary_nes'] = sf.node2domain['primary_nes'].values
    out['trim_domain_ids'] = sf.domains['id'].values.astype(np.int64)
    out['trim_domain_labels'] = np.array(list(sf.domains['label'].values))
    np.savez_compressed(os.path.join(HERE, 'domains.npz'), **out)


def make_io(safe, safe_io, nx, pd):
    """The callers / data formats either side of the path, run through the real
    reference: `.scatter` networks with their Euclidean pseudo-network (safe.py:296-309),
    `calculate_edge_lengths` on weighted edges (safe_io.py:311-333) and `read_attributes`
    (safe_io.py:336-430) for text, gzip and DataFrame inputs.  Files are stored as raw bytes."""
    import gzip
    import tempfile
    import time
    import warnings
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(77)
    out = {}
    tmp = tempfile.mkdtemp()

    # ---- .scatter network --------------------------------------------------------------------
    n = 180
    xy = clustered_layout(rng, n) * np.array([3.0, 1.0]) + np.array([-1.0, 0.25])   # x and y extents differ
    lines = ['key\tx\ty\tlabel']
    for i in range(n):
        lines.append('K%03d\t%.6f\t%.6f\tgene%d' % (i, xy[i, 0], xy[i, 1], i))
    text = ('\n'.join(lines) + '\n').encode()
    path = os.path.join(tmp, 'points.scatter')
    with open(path, 'wb') as f:
        f.write(text)
    out['scatter_file'] = np.frombuffer(text, dtype=np.uint8)
    sf = safe.SAFE(verbose=False)
    sf.neighborhood_radius = 0.07
    sf.load_network(network_file=path, node_key_attribute='key')
    out['scatter_x'] = np.array([v for _, v in sf.graph.nodes.data('x')], dtype=np.float64)
    out['scatter_y'] = np.array([v for _, v in sf.graph.nodes.data('y')], dtype=np.float64)
    out['scatter_node_key'] = np.array(list(sf.nodes['key']))
    out['scatter_node_label'] = np.array(list(sf.nodes['label']))
    out['scatter_node_id'] = np.array(list(sf.nodes['id']), dtype=np.int64)
    e = np.array(sorted((min(u, v), max(u, v)) for u, v in sf.graph_euclidean.edges()), dtype=np.int64)
    out['scatter_pseudo_edges'] = e
    out['scatter_pseudo_weights'] = np.array(sorted(set(d['weight'] for _, _, d in sf.graph_euclidean.edges(data=True))))
    # whole flow on the scatter network: euclidean neighborhoods, binary attributes, unimodality on the pseudo-network
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.07)
    out['scatter_neighborhoods'] = sf.neighborhoods.astype(np.int8)
    m = 30
    b = np.zeros((n, m))
    for j in range(m):
        c = xy[rng.integers(n)]
        d = np.sqrt(((xy - c) ** 2).sum(1))
        b[:, j] = (d < rng.uniform(0.1, 0.4)) & (rng.uniform(size=n) < 0.85)
        if j % 3 == 0:                                   # a second, distant patch: more than one component
            c2 = xy[rng.integers(n)]
            b[:, j] = np.maximum(b[:, j], np.sqrt(((xy - c2) ** 2).sum(1)) < 0.15)
    frame = pd.DataFrame(b, index=['K%03d' % i for i in range(n)], columns=['term %d' % j for j in range(m)])
    sf.load_attributes(attribute_file=frame)
    out['scatter_attributes'] = b
    sf.compute_pvalues()
    sf.define_top_attributes()
    out['scatter_nes_binary'] = sf.nes_binary
    out['scatter_top'] = sf.attributes['top'].values.astype(np.int8)
    out['scatter_num_cc'] = sf.attributes['num_connected_components'].values.astype(np.int64)
    out['scatter_num_enriched'] = sf.attributes['num_neighborhoods_enriched'].values.astype(np.float64)

    # ---- calculate_edge_lengths with edge weights (distance x weight; weight 0 -> no length) -----
    nw = 90
    xyw = clustered_layout(rng, nw)
    eu, ev = radius_graph_edges(xyw, 0.12, rng)
    w = rng.choice([0.0, 1.0, 2.5, 0.3], size=eu.size, p=[0.1, 0.5, 0.2, 0.2])
    g = nx.Graph()
    for i in range(nw):
        g.add_node(i, x=float(xyw[i, 0]), y=float(xyw[i, 1]), label='n%d' % i, label_orf='ORF%d' % i)
    for u, v, ww in zip(eu, ev, w):
        g.add_edge(int(u), int(v), weight=float(ww))
    g.add_edge(5, 5, weight=1.0)                          # a self loop: length 0
    g = safe_io.calculate_edge_lengths(g, verbose=False)
    out['wl_xy'] = xyw
    out['wl_edge_u'] = np.append(eu, 5)
    out['wl_edge_v'] = np.append(ev, 5)
    out['wl_weight'] = np.append(w, 1.0)
    out['wl_length'] = np.array([g.edges[int(u), int(v)].get('length', np.nan)
                                 for u, v in zip(out['wl_edge_u'], out['wl_edge_v'])], dtype=np.float64)

    # ---- read_attributes ------------------------------------------------------------------------
    node_order = ['K%03d' % i for i in range(n)]
    node_order[17] = node_order[3]                        # two network nodes carrying the same key
    node_order[101] = node_order[100]
    out['ra_node_order'] = np.array(node_order)

    def file_case(tag, header, rows, gz):
        body = ('\n'.join(['\t'.join(header)] + ['\t'.join(r) for r in rows]) + '\n').encode()
        name = os.path.join(tmp, tag + ('.txt.gz' if gz else '.txt'))
        if gz:
            with gzip.GzipFile(name, 'wb', mtime=0) as f:
                f.write(body)
        else:
            with open(name, 'wb') as f:
                f.write(body)
        out[tag + '_file'] = np.frombuffer(open(name, 'rb').read(), dtype=np.uint8)
        attributes, order, mat = safe_io.read_attributes(attribute_file=name, node_label_order=list(node_order), verbose=False)
        out[tag + '_matrix'] = mat
        out[tag + '_forder'] = np.array([mat.flags['F_CONTIGUOUS'], mat.flags['C_CONTIGUOUS']])
        out[tag + '_names'] = np.array(list(attributes['name']))
        out[tag + '_ids'] = attributes['id'].values.astype(np.int64)

    # (a) GO-like binary matrix: labels = a shuffled subset of the keys + labels that are not in the network
    labels = ['K%03d' % i for i in rng.permutation(n)[:150]] + ['X%02d' % i for i in range(7)]
    ma = 12
    rows = []
    for lab in labels:
        rows.append([lab] + [str(int(v)) for v in (rng.uniform(size=ma) < 0.15)])
    file_case('ra_bin', ['ORF'] + ['GO:%07d' % j for j in range(ma)], rows, gz=False)

    # (b) quantitative, gzip: decimals that float32 cannot hold, empty cells, text garbage, duplicate labels (averaged)
    labels = ['K%03d' % i for i in rng.permutation(n)[:120]] + ['K005', 'K005', 'K040']
    mb = 9
    rows = []
    for lab in labels:
        vals = []
        for j in range(mb):
            u = rng.uniform()
            if u < 0.05:
                vals.append('')
            elif u < 0.08:
                vals.append('n/a')
            elif j < 3:
                vals.append('%.2f' % rng.choice([0.5, 0.25, -1.75, 2.0, 0.0]))       # exactly representable: down-cast to f32
            elif j < 6:
                vals.append(repr(float(rng.normal())))
            else:
                vals.append('%.4e' % (rng.normal() * 10.0 ** rng.integers(-30, 30)))
        rows.append([lab] + vals)
    file_case('ra_q', ['gene'] + ['score %d' % j for j in range(mb)], rows, gz=True)

    # (c) all columns exactly representable in float32 (pandas down-casts the whole frame), with duplicates in the file
    labels = ['K%03d' % i for i in rng.permutation(n)[:100]] + ['K011', 'K011']
    rows = [[lab] + ['%.3f' % (rng.integers(-8, 9) / 8.0) for _ in range(5)] for lab in labels]
    file_case('ra_f32', ['gene'] + ['c%d' % j for j in range(5)], rows, gz=False)

    # (d) DataFrame input, fill_value=0, mask_duplicates with a seeded global RNG
    frame = pd.DataFrame(rng.normal(size=(140, 6)), index=['K%03d' % i for i in rng.permutation(n)[:140]],
                         columns=['a', 'b', 'c', 'd', 'e', 'f'])
    out['ra_df_values'] = frame.values.copy()
    out['ra_df_index'] = np.array(list(frame.index))
    np.random.seed(3)
    attributes, order, mat = safe_io.read_attributes(attribute_file=frame.copy(), node_label_order=list(node_order),
                                                     mask_duplicates=True, fill_value=0, verbose=False)
    out['ra_df_matrix'] = mat
    out['ra_df_forder'] = np.array([mat.flags['F_CONTIGUOUS'], mat.flags['C_CONTIGUOUS']])
    out['ra_df_names'] = np.array(list(attributes['name']))
    # (e) no node order given: the file's own (sorted-unique after averaging) order
    attributes, order, mat = safe_io.read_attributes(attribute_file=frame.copy(), verbose=False)
    out['ra_df_noorder_matrix'] = mat
    out['ra_df_noorder_order'] = np.array(list(order))
    np.savez_compressed(os.path.join(HERE, 'io.npz'), **out)


def _run_reference(safe, pd, g, A, mat, **kw):
    """One compute_pvalues call of the real reference on a prepared network; returns its outputs."""
    import time
    attrs = {k: kw.pop(k) for k in ('attribute_sign', 'random_seed') if k in kw}
    sf = new_safe(safe, g, **attrs)
    sf.neighborhoods = A
    set_attributes(pd, sf, mat)
    real_sleep = time.sleep
    time.sleep = lambda s: None            # skip the fixed 1 s pause (safe.py:484)
    try:
        sf.compute_pvalues(verbose=False, **kw)
    finally:
        time.sleep = real_sleep
    out = {'pvalues_pos': sf.pvalues_pos, 'nes': sf.nes, 'nes_binary': sf.nes_binary,
           'num_enriched': sf.attributes['num_neighborhoods_enriched'].values.astype(np.float64)}
    if sf.pvalues_neg is not None:
        out['pvalues_neg'] = sf.pvalues_neg
        out['ns'] = sf.ns
    return out


def make_fdr(safe, safe_io, nx, pd):
    """multiple_testing=True through the UNSTUBBED reference (safe.py:30, 536-542, 599-605) with the real
    statsmodels.stats.multitest.fdrcorrection, plus fdrcorrection itself on tie-heavy rows."""
    import warnings
    import statsmodels
    from statsmodels.stats.multitest import fdrcorrection
    assert getattr(statsmodels, '__file__', None), 'statsmodels is stubbed: fdr.npz cannot be generated'
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(31337)
    out = {'statsmodels_version': np.array(statsmodels.__version__)}

    # ---- fdrcorrection on rows of every interesting length: ties, zeros, ones, multiples of 1/P --------
    lengths = (1, 2, 3, 7, 64, 129, 1000, 4373)
    out['row_lengths'] = np.array(lengths, dtype=np.int64)
    for n in lengths:
        rows = [rng.uniform(size=n),                                        # distinct values
                rng.integers(0, 41, size=n) / 40.0,                         # counts / P: heavy ties, 0 and 1 present
                np.where(rng.uniform(size=n) < 0.7, 1.0, rng.uniform(size=n) ** 4),   # mostly 1 (hypergeometric shape)
                np.full(n, 0.02)]                                           # one value
        rows.append(rows[0].copy())
        rows[-1][rng.integers(n)] = np.nan                                  # a NaN poisons the whole row
        p = np.stack(rows)
        out['rows_n%d_p' % n] = p
        out['rows_n%d_adj' % n] = np.stack([fdrcorrection(r)[1] for r in p])

    # ---- the whole call: a small clustered network, default metric -----------------------------------
    n = 150
    xy = clustered_layout(rng, n, n_blobs=4)
    eu, ev = radius_graph_edges(xy, 0.09, rng)
    g = make_graph(nx, xy, eu, ev)
    g = safe_io.calculate_edge_lengths(g, verbose=False)
    out.update({'xy': xy, 'edge_u': eu, 'edge_v': ev})
    sf = new_safe(safe, g)
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    A = sf.neighborhoods.copy()
    out['A'] = A.astype(np.int8)

    nan_rows = rng.choice(n, size=11, replace=False)

    def binary(m):
        b = (rng.uniform(size=(n, m)) < rng.uniform(0.03, 0.3, size=m)).astype(np.float64)
        b[nan_rows] = np.nan
        return b

    def quantitative(m):
        b = rng.normal(size=(n, m))
        b[nan_rows] = np.nan
        b[rng.uniform(size=(n, m)) < 0.03] = np.nan
        return b

    cases = []
    m = 21
    b_q = quantitative(m)
    b_qz = b_q.copy()
    b_qz[:, 4] = np.nan                              # an all-NaN column: z-score NaN -> NaN p-values -> whole rows NaN
    b_bin = binary(m)
    b_bin_nan = b_bin.copy()
    b_bin_nan[~np.isnan(b_bin_nan[:, 7]), 7] *= 0.5  # halves: hypergeom.sf is NaN where the hit count is not an integer
    b_bin_nan_all = b_bin_nan.copy()
    b_bin_nan_all[np.flatnonzero(b_bin_nan_all[:, 3] == 1)[0], 3] = 0.25   # a non-integer column total: NaN in every row
    assert np.nansum(b_bin_nan_all[:, 3]) % 1 != 0
    out.update({'b_q': b_q, 'b_qz': b_qz, 'b_bin': b_bin, 'b_bin_nan': b_bin_nan, 'b_bin_nan_all': b_bin_nan_all})
    seed = 100
    for sign in ('both', 'highest', 'lowest'):
        cases.append(('rnd_sum_' + sign, 'b_q', dict(how='randomization', neighborhood_score_type='sum', attribute_sign=sign,
                                                     num_permutations=40, random_seed=seed)))
        cases.append(('rnd_z_' + sign, 'b_qz', dict(how='randomization', neighborhood_score_type='z-score', attribute_sign=sign,
                                                    num_permutations=30, random_seed=seed + 1)))
        seed += 2
    cases.append(('rnd_bin', 'b_bin', dict(how='randomization', neighborhood_score_type='sum', num_permutations=50, random_seed=9)))
    cases.append(('rnd_net', 'b_q', dict(how='randomization', neighborhood_score_type='sum', background='network',
                                         num_permutations=30, random_seed=10)))
    cases.append(('hyp', 'b_bin', dict()))
    cases.append(('hyp_net', 'b_bin', dict(background='network')))
    cases.append(('hyp_nan', 'b_bin_nan', dict(how='hypergeometric')))
    cases.append(('hyp_nan_all', 'b_bin_nan_all', dict(how='hypergeometric')))
    for width in (1, 2, 129):                        # rows of length 1 / 2 / 129 attributes
        out['b_q_m%d' % width] = quantitative(width)
        out['b_bin_m%d' % width] = binary(width)
        cases.append(('rnd_m%d' % width, 'b_q_m%d' % width, dict(how='randomization', neighborhood_score_type='sum',
                                                                   num_permutations=25, random_seed=200 + width)))
        cases.append(('hyp_m%d' % width, 'b_bin_m%d' % width, dict()))
    names = []
    for tag, key, kw in cases:
        res = _run_reference(safe, pd, g, A, out[key].copy(), multiple_testing=True, **dict(kw))
        names.append(tag)
        out[tag + '_input'] = np.array(key)
        out[tag + '_kwargs'] = np.array(repr(kw))
        for k, v in res.items():
            out[tag + '_' + k] = v
    out['cases'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'fdr.npz'), **out)


def make_big(safe, safe_io, nx, pd):
    """A second size (all other vectors are N = 257, P <= 40): N = 1200 ragged clustered layout, default metric,
    300 permutations (> 255: counter carries past a byte; several 256-row groups for the matrix-core kernel;
    multi-level width classes of the blocked bit-sliced kernel).  Quantitative f64 (sum and z-score) and binary
    attributes through SAFE.compute_pvalues of the real reference (safe.py:432-554, safe_extras.py:36-70)."""
    import warnings
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(1200)
    n = 1200
    xy = clustered_layout(rng, n, n_blobs=9, spread=0.05)
    eu, ev = radius_graph_edges(xy, 0.035, rng, keep=0.7)
    g = make_graph(nx, xy, eu, ev)
    g = safe_io.calculate_edge_lengths(g, verbose=False)
    el = np.array([g.edges[int(u), int(v)]['length'] for u, v in zip(eu, ev)], dtype=np.float64)
    out = {'xy': xy, 'edge_u': eu.astype(np.int32), 'edge_v': ev.astype(np.int32), 'edge_length': el}
    sf = new_safe(safe, g)
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.15)
    A = sf.neighborhoods.copy()
    out['A_bits'] = np.packbits(A.astype(np.uint8), axis=1)
    out['A_row_counts'] = A.sum(axis=1).astype(np.int64)

    nan_rows = rng.choice(n, size=57, replace=False)
    mb = 160                                                     # 2.5 64-attribute word groups
    sizes = np.exp(rng.uniform(np.log(2), np.log(400), size=mb))  # GO-like term sizes, log-uniform
    b_bin = (rng.uniform(size=(n, mb)) < (sizes / n)).astype(np.float32)
    b_bin[nan_rows] = np.nan
    b_bin = np.asfortranarray(b_bin)                             # the .txt.gz loader's layout
    mq = 48
    b_q = rng.normal(size=(n, mq))
    b_q[:, :8] = np.round(b_q[:, :8] * 64) / 64                  # dyadic columns: exactly representable on the fixed-point grid
    b_q[nan_rows] = np.nan
    b_q[rng.uniform(size=(n, mq)) < 0.01] = np.nan
    out['b_bin'] = np.packbits(np.nan_to_num(b_bin).astype(np.uint8), axis=0)
    out['b_bin_nan_rows'] = np.sort(nan_rows).astype(np.int32)
    out['b_q'] = b_q

    nperm = 300
    for tag, mat, kw in (('bin', b_bin, dict(neighborhood_score_type='sum', random_seed=41)),
                         ('q_sum', b_q, dict(neighborhood_score_type='sum', random_seed=42)),
                         ('q_z', b_q, dict(neighborhood_score_type='z-score', random_seed=43))):
        res = _run_reference(safe, pd, g, A, mat.copy(order='K'), how='randomization', num_permutations=nperm, **kw)
        out[tag + '_meta'] = np.array([nperm, kw['random_seed']], dtype=np.int64)
        # p = counts / P exactly (safe.py:532-533): the integer counts carry the same information in 2 bytes
        for side in ('neg', 'pos'):
            p = res['pvalues_' + side]
            c = np.where(np.isnan(p), -1, np.rint(p * nperm)).astype(np.int16)
            back = np.where(c < 0, np.nan, c / float(nperm))
            assert np.array_equal(back, p, equal_nan=True)
            out[tag + '_counts_' + side] = c
        # NES takes few distinct values (a function of the two counts): dictionary-coded, nes = values[codes] bit for bit
        vals, codes = np.unique(res['nes'], return_inverse=True)
        assert vals.size < 65536 and np.array_equal(vals[codes].reshape(res['nes'].shape), res['nes'], equal_nan=True)
        out[tag + '_nes_values'] = vals
        out[tag + '_nes_codes'] = codes.reshape(res['nes'].shape).astype(np.uint16)
        out[tag + '_nes_binary'] = res['nes_binary'].astype(np.int8)
        out[tag + '_num_enriched'] = res['num_enriched']
        if tag == 'bin':
            assert np.array_equal(res['ns'], np.rint(res['ns']))
            out[tag + '_ns'] = res['ns'].astype(np.int16)
        else:
            out[tag + '_ns'] = res['ns']
    np.savez_compressed(os.path.join(HERE, 'big.npz'), **out)


def main():
    import networkx as nx
    import pandas as pd
    safe, safe_extras, safe_io = import_reference()
    rng = np.random.default_rng(20240917)
    out = {}

    # ---------------- neighborhoods -------------------------------------------------
    n = 257
    xy = clustered_layout(rng, n)
    eu, ev = radius_graph_edges(xy, 0.06, rng)
    g = make_graph(nx, xy, eu, ev)
    g = safe_io.calculate_edge_lengths(g, verbose=False)       # reference's own edge 'length'
    el = np.array([g.edges[int(u), int(v)]['length'] for u, v in zip(eu, ev)], dtype=np.float64)
    nbr = {'xy': xy, 'edge_u': eu, 'edge_v': ev, 'edge_length': el}

    for radius in (0.05, 0.15):
        sf = new_safe(safe, g)
        sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=radius)
        nbr['euclidean_r%g' % radius] = sf.neighborhoods.astype(np.int8)
    for radius in (0.08, 0.2):
        sf = new_safe(safe, g)
        sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=radius)
        nbr['swl_r%g' % radius] = sf.neighborhoods.astype(np.int8)
        dm = np.full((n, n), np.inf)
        for s, row in sf.node_distances.items():
            for t, d in row.items():
                dm[s, t] = d
        nbr['swl_dist_r%g' % radius] = dm
    for radius in (1, 2, 3):
        sf = new_safe(safe, g)
        sf.define_neighborhoods(node_distance_metric='shortpath', neighborhood_radius=radius)
        nbr['shortpath_r%d' % radius] = sf.neighborhoods.astype(np.int8)
    # the full distance matrix the euclidean branch thresholds (scipy pdist, safe.py:397)
    from scipy.spatial.distance import pdist, squareform
    nbr['euclidean_dist'] = squareform(pdist(xy, 'euclidean'))
    np.savez_compressed(os.path.join(HERE, 'neighborhoods.npz'), **nbr)

    # shared neighborhood structure for the enrichment cases (default metric)
    sf = new_safe(safe, g)
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.2)
    A = sf.neighborhoods.copy()

    # ---------------- attributes ----------------------------------------------------
    m = 24
    nan_rows = rng.choice(n, size=19, replace=False)
    b_bin = (rng.uniform(size=(n, m)) < rng.uniform(0.01, 0.2, size=m)).astype(np.float64)
    b_bin[nan_rows] = np.nan
    b_bin[:, 5] = np.nan                       # an all-NaN attribute
    b_bin[nan_rows[:3], 5] = np.nan
    b_bin_f32F = np.asfortranarray(b_bin.astype(np.float32))      # .txt.gz loader layout (safe_io.py:361,410)

    b_q = rng.normal(size=(n, m))
    b_q[nan_rows] = np.nan
    b_q[rng.uniform(size=(n, m)) < 0.03] = np.nan
    b_q[:, 2] = np.where(rng.uniform(size=n) < 0.9, 0.0, b_q[:, 2])    # sparse column with many ties
    b_q_f32 = b_q.astype(np.float32)

    b_int = rng.integers(0, 4, size=(n, m)).astype(np.float64)        # small integers, not binary
    b_int[nan_rows] = np.nan

    enr = {'A': A.astype(np.int8), 'b_bin': b_bin, 'b_q': b_q, 'b_q_f32': b_q_f32, 'b_int': b_int}

    # hypergeometric (auto dispatch on binary data) for f64-C and f32-F inputs, both backgrounds
    for tag, mat, bg in (('hyp_f64', b_bin.copy(), 'attribute_file'),
                         ('hyp_f32F', b_bin_f32F.copy(order='F'), 'attribute_file'),
                         ('hyp_net', b_bin.copy(), 'network')):
        sf = new_safe(safe, g)
        sf.neighborhoods = A
        set_attributes(pd, sf, mat)
        sf.compute_pvalues(background=bg)
        assert sf.pvalues_neg is None
        enr[tag + '_pvalues_pos'] = sf.pvalues_pos
        enr[tag + '_nes'] = sf.nes
        enr[tag + '_nes_binary'] = sf.nes_binary
        enr[tag + '_num_enriched'] = sf.attributes['num_neighborhoods_enriched'].values.astype(np.float64)

    # hypergeometric forced on small-integer (non-binary) data
    sf = new_safe(safe, g)
    sf.neighborhoods = A
    set_attributes(pd, sf, b_int.copy())
    sf.compute_pvalues(how='hypergeometric')
    enr['hyp_int_pvalues_pos'] = sf.pvalues_pos
    enr['hyp_int_nes'] = sf.nes

    # randomization
    cases = [
        ('rnd_bin_sum', b_bin_f32F.copy(order='F'), 'sum', 'both', 'attribute_file', 40, 7),
        ('rnd_q_sum', b_q.copy(), 'sum', 'both', 'attribute_file', 40, 11),
        ('rnd_q32_sum_hi', b_q_f32.copy(), 'sum', 'highest', 'attribute_file', 30, 3),
        ('rnd_q_sum_lo', b_q.copy(), 'sum', 'lowest', 'attribute_file', 30, 5),
        ('rnd_q_z', b_q.copy(), 'z-score', 'both', 'attribute_file', 30, 13),
        ('rnd_q32_z', b_q_f32.copy(), 'z-score', 'both', 'attribute_file', 20, 17),
        ('rnd_q_net', b_q.copy(), 'sum', 'both', 'network', 30, 19),
        ('rnd_int_sum', b_int.copy(), 'sum', 'both', 'attribute_file', 30, 23),
    ]
    import time
    real_sleep = time.sleep
    time.sleep = lambda s: None            # skip the fixed 1 s pause (safe.py:484)
    for tag, mat, score, sign, bg, nperm, seed in cases:
        sf = new_safe(safe, g, attribute_sign=sign, random_seed=seed)
        sf.neighborhoods = A
        set_attributes(pd, sf, mat)
        sf.compute_pvalues(how='randomization', neighborhood_score_type=score, background=bg,
                           num_permutations=nperm, verbose=False)
        enr[tag + '_meta'] = np.array([nperm, seed], dtype=np.int64)
        enr[tag + '_ns'] = sf.ns
        enr[tag + '_pvalues_neg'] = sf.pvalues_neg
        enr[tag + '_pvalues_pos'] = sf.pvalues_pos
        enr[tag + '_nes'] = sf.nes
        enr[tag + '_nes_binary'] = sf.nes_binary
        enr[tag + '_num_enriched'] = sf.attributes['num_neighborhoods_enriched'].values.astype(np.float64)
    time.sleep = real_sleep

    # the module-level functions directly (safe_extras.py:6, :36)
    enr['score_sum_q'] = safe_extras.compute_neighborhood_score(A, b_q, 'sum')
    enr['score_z_q'] = safe_extras.compute_neighborhood_score(A, b_q, 'z-score')
    enr['score_z_q32'] = safe_extras.compute_neighborhood_score(A, b_q_f32, 'z-score')
    enr['score_sum_binF'] = safe_extras.compute_neighborhood_score(A, b_bin_f32F, 'sum')
    cn, cp = safe_extras.run_permutations((A, b_q, 'sum', 25, 29), verbose=False)
    enr['runperm_q_neg'] = cn
    enr['runperm_q_pos'] = cp
    cn, cp = safe_extras.run_permutations((A, b_bin, 'sum', 25, 31), verbose=False)
    enr['runperm_bin_neg'] = cn
    enr['runperm_bin_pos'] = cp
    np.savez_compressed(os.path.join(HERE, 'enrichment.npz'), **enr)

    # ---------------- RNG known answers (legacy np.random, safe_extras.py:46,58) -----
    kat = {}
    for seed in (0, 42, 12345, 4294967295):
        np.random.seed(seed)
        for n_items in (1, 2, 10, 257, 3971):
            base = np.arange(n_items) * 3 + 1
            kat['s%d_n%d_a' % (seed, n_items)] = np.random.permutation(base)
            kat['s%d_n%d_b' % (seed, n_items)] = np.random.permutation(base)
    np.savez_compressed(os.path.join(HERE, 'rng_kat.npz'), **kat)

    make_domains(safe, safe_io, nx, pd)
    make_io(safe, safe_io, nx, pd)
    make_fdr(safe, safe_io, nx, pd)
    make_big(safe, safe_io, nx, pd)

    for f in ('neighborhoods.npz', 'enrichment.npz', 'rng_kat.npz', 'domains.npz', 'io.npz', 'fdr.npz', 'big.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] in ('io', 'fdr', 'big'):   # only one file (the others stay as committed)
        import networkx
        import pandas
        _safe, _extras, _io = import_reference()
        {'io': make_io, 'fdr': make_fdr, 'big': make_big}[sys.argv[1]](_safe, _io, networkx, pandas)
    else:
        main()
