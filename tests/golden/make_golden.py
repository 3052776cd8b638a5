#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference.

Run in the build container only (needs /root/reference, networkx, pandas):

    python tests/golden/make_golden.py

It imports baryshnikova-lab/safepy from /root/reference (statsmodels, which the
image lacks and the hot path never calls with multiple_testing=False, is stubbed
in sys.modules), pushes seeded synthetic inputs through
``SAFE.define_neighborhoods`` / ``SAFE.compute_pvalues`` /
``safe_extras.run_permutations`` and stores inputs + outputs as small ``.npz``
files next to this script.  Only data is written: no reference source travels.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def import_reference():
    sm = types.ModuleType('statsmodels')
    sms = types.ModuleType('statsmodels.stats')
    smm = types.ModuleType('statsmodels.stats.multitest')

    def fdrcorrection(*a, **k):
        raise RuntimeError('statsmodels is stubbed; multiple_testing is out of scope')
    smm.fdrcorrection = fdrcorrection
    sm.stats = sms
    sms.multitest = smm
    sys.modules.setdefault('statsmodels', sm)
    sys.modules.setdefault('statsmodels.stats', sms)
    sys.modules.setdefault('statsmodels.stats.multitest', smm)
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

    for f in ('neighborhoods.npz', 'enrichment.npz', 'rng_kat.npz', 'domains.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
