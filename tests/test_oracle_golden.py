"""Pin the CPU oracle against vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import ast

import numpy as np
import pytest

from oracle import safe_oracle as orc

RADII_E = (0.05, 0.15)
RADII_W = (0.08, 0.2)


def test_euclidean_masks_bit_exact(golden_nbr):
    g = golden_nbr
    for r in RADII_E:
        got = orc.neighborhoods_euclidean(g['xy'], r)
        assert got.dtype == np.int64
        assert np.array_equal(got, g['euclidean_r%g' % r].astype(np.int64))
        assert np.all(np.diag(got) == 1)
    assert np.array_equal(orc.euclidean_distances(g['xy']), g['euclidean_dist'])


def test_edge_lengths_match_reference(golden_nbr):
    g = golden_nbr
    assert np.array_equal(orc.edge_lengths(g['xy'], g['edge_u'], g['edge_v']), g['edge_length'])


def test_weighted_shortpath_masks_and_distances(golden_nbr):
    g = golden_nbr
    n = g['xy'].shape[0]
    for r in RADII_W:
        cutoff = orc.layout_radius(g['xy'][:, 0], r)
        a, d = orc.neighborhoods_shortpath(n, g['edge_u'], g['edge_v'], g['edge_length'], cutoff)
        assert np.array_equal(a, g['swl_r%g' % r].astype(np.int64))
        assert np.array_equal(d, g['swl_dist_r%g' % r])       # bit-exact f64 path lengths


def test_unweighted_shortpath_masks(golden_nbr):
    g = golden_nbr
    n = g['xy'].shape[0]
    ones = np.ones(g['edge_u'].shape[0])
    for r in (1, 2, 3):
        a, _ = orc.neighborhoods_shortpath(n, g['edge_u'], g['edge_v'], ones, r)
        assert np.array_equal(a, g['shortpath_r%d' % r].astype(np.int64))


def test_scores(golden_enr):
    g = golden_enr
    a = g['A'].astype(np.int64)
    for key, mat, kind in (('score_sum_q', g['b_q'], 'sum'), ('score_z_q', g['b_q'], 'z-score'),
                           ('score_z_q32', g['b_q_f32'], 'z-score')):
        np.testing.assert_allclose(orc.compute_neighborhood_score(a, mat, kind), g[key],
                                   rtol=1e-12, atol=0, equal_nan=True)


def test_run_permutations(golden_enr):
    g = golden_enr
    a = g['A'].astype(np.int64)
    cn, cp = orc.run_permutations(a, g['b_q'], 'sum', 25, 29)
    assert np.array_equal(cn, g['runperm_q_neg']) and np.array_equal(cp, g['runperm_q_pos'])
    cn, cp = orc.run_permutations(a, g['b_bin'], 'sum', 25, 31)
    assert np.array_equal(cn, g['runperm_bin_neg']) and np.array_equal(cp, g['runperm_bin_pos'])


def test_index_table_reproduces_row_permutation(golden_enr):
    g = golden_enr
    a = g['A'].astype(np.int64)
    b = g['b_q']
    table = orc.permutation_index_table(b, 25, 29)
    obs = orc.compute_neighborhood_score(a, b, 'sum')
    cn = np.zeros(obs.shape)
    cp = np.zeros(obs.shape)
    for k in range(25):
        s = orc.compute_neighborhood_score(a, b[table[k]], 'sum')
        cn += s <= obs
        cp += s >= obs
    assert np.array_equal(cn, g['runperm_q_neg']) and np.array_equal(cp, g['runperm_q_pos'])


HYP = (('hyp_f64', 'b_bin', None, 'attribute_file'), ('hyp_f32F', 'b_bin', np.float32, 'attribute_file'),
       ('hyp_net', 'b_bin', None, 'network'))


@pytest.mark.parametrize('tag,src,dtype,bg', HYP)
def test_hypergeometric(golden_enr, tag, src, dtype, bg):
    g = golden_enr
    a = g['A'].astype(np.int64)
    b = g[src].copy()
    if dtype is not None:
        b = np.asfortranarray(b.astype(dtype))
    out = orc.compute_pvalues(a, b, background=bg)
    assert 'pvalues_neg' not in out and 'ns' not in out
    np.testing.assert_array_equal(out['pvalues_pos'], g[tag + '_pvalues_pos'])
    np.testing.assert_array_equal(out['nes'], g[tag + '_nes'])
    np.testing.assert_array_equal(out['nes_binary'], g[tag + '_nes_binary'])
    np.testing.assert_array_equal(out['num_neighborhoods_enriched'], g[tag + '_num_enriched'])


def test_hypergeometric_forced_on_integers(golden_enr):
    g = golden_enr
    out = orc.compute_pvalues(g['A'].astype(np.int64), g['b_int'].copy(), enrichment_type='hypergeometric')
    np.testing.assert_array_equal(out['pvalues_pos'], g['hyp_int_pvalues_pos'])
    np.testing.assert_array_equal(out['nes'], g['hyp_int_nes'])


RND = (('rnd_bin_sum', 'b_bin', np.float32, 'sum', 'both', 'attribute_file'),
       ('rnd_q_sum', 'b_q', None, 'sum', 'both', 'attribute_file'),
       ('rnd_q32_sum_hi', 'b_q_f32', None, 'sum', 'highest', 'attribute_file'),
       ('rnd_q_sum_lo', 'b_q', None, 'sum', 'lowest', 'attribute_file'),
       ('rnd_q_z', 'b_q', None, 'z-score', 'both', 'attribute_file'),
       ('rnd_q32_z', 'b_q_f32', None, 'z-score', 'both', 'attribute_file'),
       ('rnd_q_net', 'b_q', None, 'sum', 'both', 'network'),
       ('rnd_int_sum', 'b_int', None, 'sum', 'both', 'attribute_file'))


@pytest.mark.parametrize('tag,src,dtype,score,sign,bg', RND)
def test_randomization(golden_enr, tag, src, dtype, score, sign, bg):
    g = golden_enr
    a = g['A'].astype(np.int64)
    b = g[src].copy()
    if dtype is not None:
        b = np.asfortranarray(b.astype(dtype))
    nperm, seed = (int(v) for v in g[tag + '_meta'])
    out = orc.compute_pvalues(a, b, enrichment_type='randomization', neighborhood_score_type=score,
                              background=bg, num_permutations=nperm, random_seed=seed, attribute_sign=sign)
    np.testing.assert_allclose(out['ns'], g[tag + '_ns'], rtol=1e-12, atol=0, equal_nan=True)
    for key in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        np.testing.assert_array_equal(out[key], g[tag + '_' + key])
    np.testing.assert_array_equal(out['num_neighborhoods_enriched'], g[tag + '_num_enriched'])


def test_restated_mt19937_matches_reference_stream(golden_rng):
    g = golden_rng
    for seed in (0, 42, 12345, 4294967295):
        rng = orc.LegacyMT19937(seed)
        for n_items in (1, 2, 10, 257, 3971):
            base = [3 * i + 1 for i in range(n_items)]
            for suffix in ('a', 'b'):
                want = g['s%d_n%d_%s' % (seed, n_items, suffix)]
                assert rng.permutation(base) == [int(v) for v in want]


def test_fdrcorrection_equals_statsmodels_rows(golden_fdr):
    """oracle.fdrcorrection == statsmodels.stats.multitest.fdrcorrection(row)[1] (safe.py:30, 538, 541, 604) bit for
    bit on rows of length 1 ... 4373: distinct values, counts / P with heavy ties, mostly-one rows, one value, a NaN."""
    g = golden_fdr
    assert str(g['statsmodels_version']) == '0.12.2'          # the reference pins 0.14.4: version skew, stated
    for n in g['row_lengths']:
        p, want = g['rows_n%d_p' % n], g['rows_n%d_adj' % n]
        for row, adj in zip(p, want):
            np.testing.assert_array_equal(orc.fdrcorrection(row), adj)
        np.testing.assert_array_equal(orc.fdr_rows(p), want)
    assert np.isnan(g['rows_n7_adj'][-1]).all()               # a NaN poisons the whole row


def test_compute_pvalues_with_multiple_testing_equals_reference(golden_fdr):
    """oracle.compute_pvalues(..., multiple_testing=True) == the unstubbed reference with the real statsmodels:
    adjusted p-values, NES, nes_binary and the per-attribute counts, bit for bit, every case of fdr.npz."""
    g = golden_fdr
    a = g['A'].astype(np.int64)
    assert len(g['cases']) == 18
    for tag in g['cases']:
        kw = ast.literal_eval(str(g[tag + "_kwargs"]))                     # a dict literal written by make_golden.py
        mat = g[str(g[tag + '_input'])].copy()
        got = orc.compute_pvalues(a, mat, enrichment_type=kw.get('how', 'auto'),
                                  neighborhood_score_type=kw.get('neighborhood_score_type', 'sum'),
                                  background=kw.get('background', 'attribute_file'),
                                  num_permutations=kw.get('num_permutations', 1000), random_seed=kw.get('random_seed'),
                                  attribute_sign=kw.get('attribute_sign', 'both'), multiple_testing=True)
        keys = ['pvalues_pos', 'nes', 'nes_binary'] + (['pvalues_neg'] if tag.startswith('rnd') else [])
        for k in keys:
            np.testing.assert_array_equal(got[k], g[tag + '_' + k], err_msg='%s %s' % (tag, k))
        np.testing.assert_array_equal(got['num_neighborhoods_enriched'], g[tag + '_num_enriched'])
    nan_rows = np.isnan(g['hyp_nan_pvalues_pos']).all(axis=1)    # half-integer hit counts: NaN, and it spreads over those rows
    assert 0 < nan_rows.sum() < len(nan_rows) and not np.isnan(g['hyp_nan_pvalues_pos'][~nan_rows]).any()
    assert np.isnan(g['hyp_nan_all_pvalues_pos']).all()         # a non-integer column total: every row
    assert np.isnan(g['rnd_z_both_pvalues_pos']).all() and not np.isnan(g['rnd_sum_both_pvalues_pos']).any()


def test_bh_restatement_against_scipy_independent_implementation():
    """A second, independent check: SciPy's Benjamini-Hochberg (p * n / rank instead of p / (rank / n)) equals the
    restatement to the last ulp or two."""
    import numpy as np
    from scipy.stats import false_discovery_control
    from oracle import safe_oracle as orc
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 100, 4373):
        p = rng.uniform(size=n)
        k = rng.integers(0, n, size=n // 3)
        p[k] = rng.integers(0, 5, size=n // 3) / 4.0              # ties, zeros, ones
        got = orc.fdrcorrection(p)
        np.testing.assert_allclose(got, false_discovery_control(p, method='bh'), rtol=4e-16, atol=0)
        assert got.max() <= 1 and np.all(got >= p)
    assert np.isnan(orc.fdrcorrection(np.array([0.1, np.nan, 0.5]))).all()
    rows = orc.fdr_rows(np.array([[0.01, 0.04, 0.03], [1.0, 0.0, 0.5]]))
    np.testing.assert_allclose(rows, [[0.03, 0.04, 0.04], [1.0, 0.0, 0.75]])


def test_oracle_at_the_second_size_equals_reference(golden_big):
    """N = 1200, 300 permutations (tests/golden/big.npz, the real reference): binary f32-F, quantitative f64 sum and
    z-score -- p-values, NES, nes_binary, per-attribute counts exact; scores to 1e-12."""
    g = golden_big
    for tag, mat, score in (('bin', g['b_bin'], 'sum'), ('q_sum', g['b_q'], 'sum'), ('q_z', g['b_q'], 'z-score')):
        got = orc.compute_pvalues(g['A'], mat.copy(order='K'), enrichment_type='randomization', neighborhood_score_type=score,
                                  num_permutations=g[tag + '_nperm'], random_seed=g[tag + '_seed'])
        for k in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
            np.testing.assert_array_equal(got[k], g[tag + '_' + k], err_msg='%s %s' % (tag, k))
        np.testing.assert_array_equal(got['num_neighborhoods_enriched'], g[tag + '_num_enriched'])
        np.testing.assert_allclose(got['ns'], g[tag + '_ns'], rtol=1e-12, atol=0, equal_nan=True)
    assert g['bin_pvalues_pos'].max() == 1.0 and (g['bin_pvalues_neg'] * 300 > 255).any()    # counters past one byte


def test_oracle_top_attributes_and_domains_vs_reference():
    """define_top_attributes / define_domains (safe.py:610-705) restated in the oracle vs the real
    reference's outputs (tests/golden/domains.npz)."""
    import os
    import numpy as np
    from oracle import safe_oracle as orc
    g = dict(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'domains.npz')))
    n = g['xy'].shape[0]
    top = orc.top_attributes(g['nes_binary'], g['num_enriched'], n, g['edge_u'], g['edge_v'], min_size=10)
    assert np.array_equal(top['top'], g['top'].astype(bool))
    assert np.array_equal(top['num_connected_components'], g['num_cc'])
    assert np.array_equal(top['num_large_connected_components'], g['num_large_cc'])
    for j, s in enumerate(top['size_connected_components']):
        want = g['cc_sizes'][j][g['cc_sizes'][j] >= 0]
        assert (s is None and len(want) == 0) or np.array_equal(s, want)
    for thr in (0.75, 0.65):
        tag = 'thr%g_' % thr
        dom, ids, sums, primary, primary_nes = orc.domains(g['nes'], g['nes_binary'], g['top'].astype(bool), 'jaccard', thr)
        assert np.array_equal(dom, g[tag + 'domain']) and np.array_equal(ids, g[tag + 'domain_ids'])
        assert np.array_equal(sums, g[tag + 'node2domain']) and np.array_equal(primary, g[tag + 'primary_domain'])
        np.testing.assert_array_equal(primary_nes, g[tag + 'primary_nes'])
