"""HIP path (through the C ABI) against the round-3 reference vectors: `multiple_testing=True` run through the UNSTUBBED
reference with the real statsmodels (tests/golden/fdr.npz) and the second size, N = 1200 x 300 permutations
(tests/golden/big.npz) -- every permutation-kernel family against the reference ITSELF, not only the pinned oracle."""
import ast

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


def _safe(amd, g, **attrs):
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(g['xy'], g['edge_u'], g['edge_v'])
    for k, v in attrs.items():
        setattr(sf, k, v)
    return sf


FDR_CASES = ['rnd_sum_both', 'rnd_z_both', 'rnd_sum_highest', 'rnd_z_highest', 'rnd_sum_lowest', 'rnd_z_lowest', 'rnd_bin',
             'rnd_net', 'hyp', 'hyp_net', 'hyp_nan', 'hyp_nan_all', 'rnd_m1', 'hyp_m1', 'rnd_m2', 'hyp_m2', 'rnd_m129', 'hyp_m129']


def test_fdr_fixture_lists_the_cases_tested_here(golden_fdr):
    assert sorted(str(c) for c in golden_fdr['cases']) == sorted(FDR_CASES)


@pytest.mark.parametrize('sort', ['auto', 'block', 'cub'])
@pytest.mark.parametrize('tag', FDR_CASES)
def test_multiple_testing_vs_reference(amd, golden_fdr, monkeypatch, tag, sort):
    """SAFE.compute_pvalues(multiple_testing=True) == the reference with statsmodels' fdrcorrection (safe.py:536-554,
    599-608): adjusted empirical p-values bit for bit (same divisions in the same order), hypergeometric ones within
    1e-6 relative of SciPy's tail; NES within 1e-12 (the device's log10); nes_binary and the per-attribute counts exact."""
    if sort != 'auto':                                           # 'auto': the sort-free histogram form where p = counts / P
        monkeypatch.setenv('SAFE_HIP_FDR_SORT', sort)
    g = golden_fdr
    kw = ast.literal_eval(str(g[tag + "_kwargs"]))                           # a dict literal written by make_golden.py
    attrs = {k: kw.pop(k) for k in ('attribute_sign', 'random_seed') if k in kw}
    sf = _safe(amd, g, **attrs)
    sf.neighborhoods = g['A'].astype(np.int64)
    sf.load_attributes(attribute_file=g[str(g[tag + '_input'])].copy())
    sf.compute_pvalues(multiple_testing=True, verbose=False, **kw)
    want_p = g[tag + '_pvalues_pos']
    assert np.array_equal(np.isnan(sf.pvalues_pos), np.isnan(want_p))
    if tag.startswith('rnd'):
        np.testing.assert_array_equal(sf.pvalues_pos, want_p)
        np.testing.assert_array_equal(sf.pvalues_neg, g[tag + '_pvalues_neg'])
        np.testing.assert_allclose(sf.nes, g[tag + '_nes'], rtol=1e-12, atol=1e-12, equal_nan=True)
    else:
        assert sf.pvalues_neg is None
        np.testing.assert_allclose(sf.pvalues_pos, want_p, rtol=1e-6, atol=1e-300, equal_nan=True)
        np.testing.assert_allclose(sf.nes, g[tag + '_nes'], rtol=1e-6, atol=1e-9, equal_nan=True)
    np.testing.assert_array_equal(sf.nes_binary, g[tag + '_nes_binary'])
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, g[tag + '_num_enriched'])


@pytest.mark.parametrize('sort', ['block', 'cub'])
def test_fdr_rows_equal_statsmodels(amd, golden_fdr, monkeypatch, sort):
    """safe_fdr_adjust on the rows statsmodels.stats.multitest.fdrcorrection itself adjusted (lengths 1 ... 4373; distinct
    values, counts / P with heavy ties, mostly ones, one value, a NaN): bit for bit."""
    import torch
    from safepy_amd import backend as be
    if sort != 'auto':                                           # 'auto': the sort-free histogram form where p = counts / P
        monkeypatch.setenv('SAFE_HIP_FDR_SORT', sort)
    ctx = amd.Context.default(0)
    g = golden_fdr
    for m in (int(v) for v in g['row_lengths']):
        p, want = g['rows_n%d_p' % m], g['rows_n%d_adj' % m]
        n = p.shape[0]
        t = [torch.from_numpy(p.copy()).to('cuda'), torch.empty((n, m), dtype=torch.float64, device='cuda'),
             torch.empty((n, m), dtype=torch.float64, device='cuda'), torch.empty((m,), dtype=torch.float64, device='cuda')]
        torch.cuda.synchronize()
        be.fdr_adjust(ctx, n, m, 0, 'both', 0.05, [None] + [x.data_ptr() for x in t])
        ctx.sync()
        np.testing.assert_array_equal(t[0].cpu().numpy(), want, err_msg='row length %d' % m)


# ------------------------------------------------------------------------------- the second size --------

BIG = [('bin', None, None), ('bin', 'bits', None), ('bin', 'bits', 'pre'), ('bin', 'bits', 'barrier'), ('bin', 'scatter', None),
       ('bin', 'gather', None),
       ('q_sum', None, None), ('q_sum', 'mfma', None), ('q_sum', 'lds', None), ('q_sum', 'gather', None),
       ('q_z', None, None), ('q_z', 'mfma', None), ('q_z', 'lds', None), ('q_z', 'gather', None)]


@pytest.mark.parametrize('tag,path,kernel', BIG)
def test_second_size_vs_reference(amd, golden_big, monkeypatch, tag, path, kernel):
    """N = 1200 (neighborhoods of 1 ... 325 members: several width classes of the blocked bit-sliced kernel, five 256-row
    groups of the matrix-core kernel), 300 permutations (counters carry past 255), the real reference's outputs: empirical
    p-values, NES, nes_binary and per-attribute counts EXACTLY equal for every kernel family; scores to 1e-9."""
    if path:
        monkeypatch.setenv('SAFE_HIP_FORCE_PATH', path)
    if kernel == 'pre':
        monkeypatch.setenv('SAFE_HIP_BITS_KERNEL', 'pre')
    elif kernel == 'barrier':
        monkeypatch.setenv('SAFE_HIP_BITS_PRE', '0')
    g = golden_big
    mat = (g['b_bin'] if tag == 'bin' else g['b_q']).copy(order='K')
    sf = _safe(amd, g, random_seed=g[tag + '_seed'])
    sf.neighborhoods = g['A']
    sf.load_attributes(attribute_file=mat)
    sf.compute_pvalues(how='randomization', neighborhood_score_type='z-score' if tag == 'q_z' else 'sum',
                       num_permutations=g[tag + '_nperm'], verbose=False)
    name = amd.Context.default(0).last_kernel()[0]
    if path:
        assert name.startswith('k_permtest_' + path), name
    if tag == 'bin':
        np.testing.assert_array_equal(sf.ns, g['bin_ns'])
    else:
        np.testing.assert_allclose(sf.ns, g[tag + '_ns'], rtol=1e-9, atol=1e-12, equal_nan=True)
    np.testing.assert_array_equal(sf.pvalues_neg, g[tag + '_pvalues_neg'])
    np.testing.assert_array_equal(sf.pvalues_pos, g[tag + '_pvalues_pos'])
    np.testing.assert_array_equal(sf.nes, g[tag + '_nes'])
    np.testing.assert_array_equal(sf.nes_binary, g[tag + '_nes_binary'])
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, g[tag + '_num_enriched'])


def test_second_size_neighborhoods_vs_reference(amd, golden_big):
    """define_neighborhoods with the default metric at N = 1200 (safe.py:401-417) == the reference's membership."""
    g = golden_big
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(g['xy'], g['edge_u'], g['edge_v'], length=g['edge_length'])
    sf.define_neighborhoods(node_distance_metric='shortpath_weighted_layout', neighborhood_radius=0.15)
    assert np.array_equal(sf.neighborhoods, g['A'])
    ctx = amd.Context.default(0)                                  # the edge lengths themselves (safe_io.py:311-333)
    assert np.array_equal(ctx.edge_lengths(g['xy'], g['edge_u'], g['edge_v']), g['edge_length'])
