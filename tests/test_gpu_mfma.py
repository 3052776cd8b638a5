"""Parity of the matrix-core (i8 MFMA, exact fixed point) permutation kernel against the CPU
oracle: quantitative attributes, every layout of the inputs, every way a membership handle is
made, ragged sizes, column shards, and the decline-and-fall-back rule.  Needs an MI355X."""
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


@pytest.fixture(autouse=True)
def narrow_blocks_stay_on_the_matrix_cores(monkeypatch):
    """These tests exercise the matrix-core kernel at small sizes; the product sends such narrow blocks (n x columns <= 2e5)
    to the LDS-resident f64 kernel (enrich.hip narrow_block_prefers_lds), which tests/test_gpu_example3.py and the f64-kernel
    tests cover."""
    monkeypatch.setenv('SAFE_HIP_NARROW_LDS', '0')


def _quant(rng, n, m, dtype=np.float64, order='C', nan_rows=0, nan_frac=0.0):
    b = rng.normal(size=(n, m)).astype(dtype)
    if nan_frac:
        b[rng.uniform(size=(n, m)) < nan_frac] = np.nan
    if nan_rows:
        b[rng.choice(n, nan_rows, replace=False)] = np.nan
    return np.asfortranarray(b) if order == 'F' else np.ascontiguousarray(b)


def _counts(amd, ctx, nbr, b, nperm, seed, col0=0, col1=None, flags=None, score='sum'):
    from safepy_amd import backend as be
    attr = be.Attributes.from_host(ctx, b)
    if flags is not None:
        attr.set_row_flags(flags)
    n, m = b.shape
    col1 = m if col1 is None else col1
    perms = be.Permutations(ctx, n, attr.row_flags() if flags is None else flags, nperm, seed)
    ns, neg, pos = (ctx.alloc_f64(n, col1 - col0) for _ in range(3))
    be.permtest_counts(ctx, nbr, attr, perms, score, ns.ptr, neg.ptr, pos.ptr, col0, col1)
    name = ctx.last_kernel()[0]
    out = (ns.download((n, col1 - col0)), neg.download((n, col1 - col0)), pos.download((n, col1 - col0)), name)
    perms.close()
    attr.close()
    return out


@pytest.mark.parametrize('n,m,dtype,order', [(600, 70, np.float64, 'C'), (1000, 33, np.float32, 'F'),
                                             (257, 5, np.float64, 'F'), (1300, 64, np.float32, 'C')])
def test_quantitative_counts_exact_vs_oracle(amd, ctx, n, m, dtype, order):
    rng = np.random.default_rng(n + m)
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.09)
    b = _quant(rng, n, m, dtype, order, nan_rows=n // 20, nan_frac=0.01)
    nperm, seed = 40, 11
    ns_w = orc.compute_neighborhood_score(a, b, 'sum')
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.09))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma'
    np.testing.assert_allclose(ns, ns_w, rtol=1e-9, atol=1e-12)      # north-star tolerance: 1e-6 relative
    np.testing.assert_array_equal(cn, cn_w)                            # every <= / >= decision identical
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_full_pipeline_quantitative_vs_oracle(amd):
    """SAFE.compute_pvalues on real-valued data through the MFMA kernel: p-values, NES and the
    binarised map equal the oracle's exactly (counts are integers; NES uses the caller's log10)."""
    rng = np.random.default_rng(5)
    n, m, nperm = 900, 40, 60
    xy = rng.uniform(size=(n, 2))
    b = _quant(rng, n, m, np.float64, 'C', nan_rows=30, nan_frac=0.02)
    a = orc.neighborhoods_euclidean(xy, 0.1)
    for sign in ('both', 'highest', 'lowest'):
        want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=2,
                                   attribute_sign=sign)
        sf = amd.SAFE(verbose=False)
        sf.graph = amd.LayoutGraph(xy)
        sf.random_seed = 2
        sf.attribute_sign = sign
        sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
        sf.load_attributes(attribute_file=b.copy())
        sf.compute_pvalues(num_permutations=nperm, verbose=False)
        assert amd.Context.default(0).last_kernel()[0] == 'k_permtest_mfma'
        np.testing.assert_allclose(sf.ns, want['ns'], rtol=1e-9, atol=1e-12)
        for key in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
            np.testing.assert_array_equal(getattr(sf, key), want[key])


@pytest.mark.parametrize('how', ['shortpath+layout', 'shortpath', 'dense'])
def test_node_orders_do_not_change_results(amd, ctx, how):
    """Hilbert order (layout known), Cuthill-McKee order (no layout) and a user-supplied
    asymmetric dense membership all give the oracle's counts."""
    rng = np.random.default_rng(17)
    n, m, nperm, seed = 700, 48, 30, 9
    xy = rng.uniform(size=(n, 2))
    b = _quant(rng, n, m, np.float64, 'C', nan_rows=20)
    if how == 'dense':
        a = (rng.uniform(size=(n, n)) < 0.03).astype(np.int64)
        a[3, :] = 0                                              # an empty neighborhood
        a[:, 8] = 0                                              # a node nobody contains
        nbr = amd.Neighborhoods.from_dense(ctx, a)
    else:
        from scipy.spatial import cKDTree
        pairs = cKDTree(xy).query_pairs(0.06, output_type='ndarray')
        eu, ev = pairs[:, 0], pairs[:, 1]
        w = np.sqrt(((xy[eu] - xy[ev]) ** 2).sum(axis=1))
        cutoff = 0.12
        a, _ = orc.neighborhoods_shortpath(n, eu, ev, w, cutoff)
        nbr = amd.Neighborhoods.shortpath(ctx, n, eu, ev, w, cutoff)
        if how == 'shortpath+layout':
            nbr.set_layout(xy)
        assert np.array_equal(nbr.to_dense(), a)
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma'
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_binary_and_integer_attributes_through_mfma(amd, ctx, monkeypatch):
    """0/1 and small-integer attributes are exactly representable: forcing them through the
    MFMA kernel reproduces the oracle bit for bit, including the observed sums."""
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', 'mfma')
    rng = np.random.default_rng(23)
    n, m, nperm, seed = 520, 37, 25, 4
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.08)
    b = rng.integers(-3, 4, size=(n, m)).astype(np.float64)
    b[:, :10] = (rng.uniform(size=(n, 10)) < 0.05)
    b[:, 11] = 0                                                 # an all-zero attribute
    b[:, 12] = np.nan                                            # an all-NaN attribute
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.08))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma'
    np.testing.assert_array_equal(ns, orc.compute_neighborhood_score(a, b, 'sum'))
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_column_shards_and_many_permutations(amd, ctx):
    """Column ranges (attribute shards) with global row flags, and more permutations than one
    span (two launches on alternating streams)."""
    rng = np.random.default_rng(31)
    n, m, nperm, seed = 400, 100, 150, 6
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.1)
    b = _quant(rng, n, m, np.float32, 'F', nan_rows=15)
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    for c0, c1 in ((0, 100), (0, 33), (33, 97), (97, 100)):
        ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, c0, c1)
        assert name == 'k_permtest_mfma'
        np.testing.assert_array_equal(cn, cn_w[:, c0:c1])
        np.testing.assert_array_equal(cp, cp_w[:, c0:c1])
    nbr.close()


def test_wide_dynamic_range_declines_to_f64_kernels(amd, ctx):
    """A column whose values would lose bits on the fixed-point grid (1e18 next to 1e-3) makes
    the MFMA path decline; the f64 kernels take over and the result still matches."""
    rng = np.random.default_rng(41)
    n, m, nperm, seed = 300, 8, 20, 3
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.1)
    b = rng.normal(size=(n, m)) * 1e-3
    b[7, 2] = 1e18
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name != 'k_permtest_mfma'
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


@pytest.mark.parametrize('kind,want_slices', [('small-int', 2), ('levels-0-1-2', 2), ('int-1e6', 4), ('dyadic', 4), ('f32-normal', 6),
                                              ('f64-normal', 6), ('int-2e11', 6)])
def test_slice_count_follows_the_bits_the_columns_need(amd, ctx, monkeypatch, kind, want_slices):
    """The matrix-core kernel runs with 2 / 4 / 6 i8 slices by the width of the columns' exact fixed-point image
    (<= 14 / 30 / 46 bits from the lowest set bit of any value to the top bit of the column maximum); data that fits
    is summed EXACTLY, so counts and observed sums equal the oracle's bit for bit whatever the slice count."""
    from safepy_amd import backend as be
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', 'mfma')
    rng = np.random.default_rng(61)
    n, m, nperm, seed = 700, 45, 30, 8
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.08)
    if kind == 'small-int':
        b = rng.integers(-100, 101, size=(n, m)).astype(np.float64)
    elif kind == 'levels-0-1-2':
        b = rng.integers(0, 3, size=(n, m)).astype(np.float32)
    elif kind == 'int-1e6':
        b = rng.integers(-10**6, 10**6, size=(n, m)).astype(np.float64)
    elif kind == 'dyadic':
        b = rng.integers(-2**20, 2**20, size=(n, m)).astype(np.float64) / 2**12        # multiples of 2^-12 up to 2^8
    elif kind == 'f32-normal':
        b = rng.normal(size=(n, m)).astype(np.float32)
    elif kind == 'f64-normal':
        b = rng.normal(size=(n, m))
    else:
        b = rng.integers(-2 * 10**11, 2 * 10**11, size=(n, m)).astype(np.float64)
    b[rng.choice(n, 30, replace=False)] = np.nan
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.08))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma' and be.last_mfma_slices(ctx) == want_slices
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    if kind not in ('f32-normal', 'f64-normal'):                 # exact images: the observed sums are exact integers / dyadics
        np.testing.assert_array_equal(ns, orc.compute_neighborhood_score(a, b, 'sum'))
    else:
        np.testing.assert_allclose(ns, orc.compute_neighborhood_score(a, b, 'sum'), rtol=1e-9, atol=1e-12)
    nbr.close()


@pytest.mark.parametrize('kind', ['log-normal-sigma6', 'mixed-1e-8-to-1e3', 'one-in-500-tiny'])
def test_columns_the_rounding_grid_would_hurt_go_to_the_f64_kernels(amd, ctx, kind):
    """Real-valued f64 columns are summed on a grid of 2^-46 of the column maximum.  When more than a thousandth of a
    column lies 2^20 or more below its maximum, sums made of such values would be compared at the grid's resolution
    rather than f64's: the call declines the matrix cores and the f64 kernels produce the oracle's counts."""
    rng = np.random.default_rng(71)
    n, m, nperm, seed = 400, 12, 25, 2
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.1)
    if kind == 'log-normal-sigma6':
        b = np.exp(6.0 * rng.normal(size=(n, m)))
    elif kind == 'mixed-1e-8-to-1e3':
        b = np.where(rng.uniform(size=(n, m)) < 0.3, 1e-8, 1e3) * rng.uniform(0.5, 1.5, size=(n, m))
    else:
        b = rng.normal(size=(n, m))
        b[rng.uniform(size=(n, m)) < 1 / 500] *= 1e-9
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name != 'k_permtest_mfma'
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_moderate_dynamic_range_stays_on_the_matrix_cores_and_equals_the_f64_kernel(amd, ctx, monkeypatch):
    """Log-normal columns of moderate spread (sigma 1.5: three decades) stay on the matrix cores; their <= / >= counts equal
    the f64 gather kernel's on the same seeded inputs."""
    rng = np.random.default_rng(73)
    n, m, nperm, seed = 900, 40, 40, 6
    xy = rng.uniform(size=(n, 2))
    b = np.exp(1.5 * rng.normal(size=(n, m))) * np.where(rng.uniform(size=(n, m)) < 0.5, -1.0, 1.0)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.07))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma'
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', 'gather')
    ns_g, cn_g, cp_g, name_g = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name_g.startswith('k_permtest_gather')
    np.testing.assert_array_equal(cn, cn_g)
    np.testing.assert_array_equal(cp, cp_g)
    np.testing.assert_allclose(ns, ns_g, rtol=1e-9, atol=1e-9)          # (sums on the 2^-46 grid of the column maximum)
    nbr.close()


@pytest.mark.parametrize('order,m,dtype', [('F', 205, np.float32), ('C', 205, np.float32), ('C', 208, np.float32),
                                           ('C', 205, np.float64), ('C', 206, np.float64), ('F', 206, np.float64)])
def test_binary_neighborhood_score_on_matrix_cores(amd, ctx, monkeypatch, order, m, dtype):
    """compute_neighborhood_score 'sum' of 0/1 attributes through the matrix-core count kernel
    (one i8 plane per 32-column tile): exact integer counts, ragged column count, NaN rows; bit planes
    from Fortran-order (transposing kernel) and C-order matrices (vector loads when the row pitch is a
    multiple of 16 bytes, scalar loads otherwise)."""
    monkeypatch.setenv('SAFE_HIP_COUNTS', 'mfma')
    rng = np.random.default_rng(71)
    n = 777
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.11)
    b = (rng.uniform(size=(n, m)) < 0.07).astype(dtype)
    b[rng.choice(n, 30, replace=False)] = np.nan
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.11)
    got = amd.compute_neighborhood_score(sf.neighborhoods, np.asfortranarray(b) if order == 'F' else np.ascontiguousarray(b), 'sum')
    assert ctx.last_kernel()[0] == 'k_permtest_mfma<counts>'
    np.testing.assert_array_equal(got, orc.compute_neighborhood_score(a, b, 'sum'))


# ---------------------------------------------------------------------------------------------------------------------
# z-scores on the matrix cores (safe_extras.py:19-31): sum, sum of squares and count of a neighborhood ride in one
# 32-column MFMA tile (16 attribute columns: value digits | square digits, + the not-NaN slice)
# ---------------------------------------------------------------------------------------------------------------------
def _zdata(rng, n, m, kind, dtype, order):
    if kind == 'dyadic':                  # multiples of 1/8: sums, squares and their sums are exact in f64 AND on the fixed-point grid
        b = (rng.integers(-40, 41, size=(n, m)) / 8.0).astype(dtype)
    else:
        b = rng.normal(size=(n, m)).astype(dtype)
    b[rng.uniform(size=(n, m)) < 0.02] = np.nan
    b[rng.choice(n, n // 25, replace=False)] = np.nan
    if m > 4:
        b[:, 1] = np.nan                                          # nothing to score at all
        b[:, 2] = 2.5                                             # zero variance everywhere: std == 0 -> NaN (safe_extras.py:29)
        keep = rng.choice(n, max(3, n // 40), replace=False)      # a column so sparse that most neighborhoods see < 3 values (:30)
        col = np.full(n, np.nan)
        col[keep] = b[keep, 3]
        b[:, 3] = col
    return np.asfortranarray(b) if order == 'F' else np.ascontiguousarray(b)


@pytest.mark.parametrize('n,m,dtype,order,kind', [(600, 37, np.float64, 'C', 'normal'), (1000, 16, np.float32, 'F', 'normal'),
                                                  (257, 5, np.float64, 'F', 'dyadic'), (1300, 50, np.float64, 'C', 'dyadic'),
                                                  (900, 17, np.float32, 'C', 'dyadic')])
def test_zscore_counts_on_matrix_cores_vs_oracle(amd, ctx, n, m, dtype, order, kind):
    rng = np.random.default_rng(3 * n + m)
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.09)
    b = _zdata(rng, n, m, kind, dtype, order)
    nperm, seed = 40, 11
    ns_w = orc.compute_neighborhood_score(a, b, 'z-score')
    cn_w, cp_w = orc.run_permutations(a, b, 'z-score', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.09))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name == 'k_permtest_mfma'
    assert np.array_equal(np.isnan(ns), np.isnan(ns_w))               # < 3 values, zero variance, empty: NaN in the same places
    if kind == 'dyadic':
        np.testing.assert_array_equal(ns, ns_w)                       # exact sums in, the reference's operations in its order: same bits
    else:
        np.testing.assert_allclose(ns, ns_w, rtol=1e-9, atol=1e-12)   # north-star tolerance: 1e-6 relative
    np.testing.assert_array_equal(cn, cn_w)                           # every <= / >= decision identical
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_zscore_matrix_cores_agree_with_the_f64_kernels(amd, ctx, monkeypatch):
    """Same inputs through the matrix-core form and (SAFE_HIP_MFMA_Z=0) through the f64 kernels, column shards included."""
    rng = np.random.default_rng(77)
    n, m, nperm, seed = 1500, 41, 50, 5
    xy = rng.uniform(size=(n, 2))
    b = _zdata(rng, n, m, 'normal', np.float64, 'C')
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.07))
    flags = (~np.isnan(b)).any(axis=1).astype(np.uint8)
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name == 'k_permtest_mfma'
    for c0, c1 in ((0, 16), (16, 17), (17, 41)):                      # tiles of 16 columns: aligned, single, ragged
        ns_s, cn_s, cp_s, name_s = _counts(amd, ctx, nbr, b, nperm, seed, c0, c1, flags=flags, score='z-score')
        assert name_s == 'k_permtest_mfma'
        np.testing.assert_array_equal(ns_s, ns[:, c0:c1])
        np.testing.assert_array_equal(cn_s, cn[:, c0:c1])
        np.testing.assert_array_equal(cp_s, cp[:, c0:c1])
    monkeypatch.setenv('SAFE_HIP_MFMA_Z', '0')
    ns_f, cn_f, cp_f, name_f = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name_f != 'k_permtest_mfma'
    np.testing.assert_allclose(ns, ns_f, rtol=1e-9, atol=1e-12, equal_nan=True)
    np.testing.assert_array_equal(cn, cn_f)
    np.testing.assert_array_equal(cp, cp_f)
    nbr.close()


def test_full_pipeline_zscore_vs_oracle(amd):
    """SAFE.compute_pvalues(neighborhood_score_type='z-score') on a network too large for the LDS-resident f64 kernel's
    comfort runs on the matrix cores; p-values, NES and the binarised map equal the oracle's."""
    rng = np.random.default_rng(8)
    n, m, nperm = 1100, 24, 50
    xy = rng.uniform(size=(n, 2))
    b = _zdata(rng, n, m, 'normal', np.float64, 'C')
    a = orc.neighborhoods_euclidean(xy, 0.08)
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=4,
                               neighborhood_score_type='z-score')
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.random_seed = 4
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.08)
    sf.load_attributes(attribute_file=b.copy())
    sf.compute_pvalues(num_permutations=nperm, neighborhood_score_type='z-score', verbose=False)
    assert amd.Context.default(0).last_kernel()[0] == 'k_permtest_mfma'
    np.testing.assert_allclose(sf.ns, want['ns'], rtol=1e-9, atol=1e-12, equal_nan=True)
    for key in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        np.testing.assert_array_equal(getattr(sf, key), want[key])


def test_zscore_sparse_inexact_column_is_left_to_the_f64_kernels(amd, ctx):
    """A z-score is scale-free: with one non-zero value among a neighborhood's members it is +-1/sqrt(n - 1) whatever the value,
    so a sparse column produces mathematically equal scores that the reference orders by f64 rounding.  Only the same f64
    operands reproduce that -- the matrix-core form declines such a call (exact zeros in a column it cannot hold exactly)."""
    rng = np.random.default_rng(31)
    n, m, nperm, seed = 800, 20, 40, 3
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.06)
    b = rng.normal(size=(n, m))
    b[:, 5] = np.where(rng.uniform(size=n) < 0.9, 0.0, b[:, 5])
    cn_w, cp_w = orc.run_permutations(a, b, 'z-score', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.06))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name != 'k_permtest_mfma'
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    # the same column on a grid the fixed-point image holds exactly (small integers): the matrix cores take it, same decisions
    b[:, 5] = np.where(b[:, 5] == 0.0, 0.0, np.rint(4 * b[:, 5]))
    others = [j for j in range(m) if j != 5]
    b[:, others] = np.rint(64 * b[:, others]) / 8.0
    cn_w, cp_w = orc.run_permutations(a, b, 'z-score', nperm, seed)
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name == 'k_permtest_mfma'
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


# ---------------------------------------------------------------------------------------------------------------------
# the filtered form: three of the six slices on the matrix cores, undecided compares settled from the low digits
# ---------------------------------------------------------------------------------------------------------------------
def _filter_case(seed, n=700, m=45):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.1)
    b = _quant(rng, n, m, np.float64, 'C', nan_rows=n // 25, nan_frac=0.01)
    return xy, a, b


def test_filtered_form_runs_three_slices_and_equals_the_six_slice_form_and_the_oracle(amd, ctx, monkeypatch):
    from safepy_amd import backend as be
    xy, a, b = _filter_case(101)
    nperm, seed = 120, 4
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    ns_w = orc.compute_neighborhood_score(a, b, 'sum')
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma' and be.last_mfma_slices(ctx) == 6
    core, undecided = be.last_mfma_filter(ctx)
    assert core == 3 and undecided >= 0
    np.testing.assert_allclose(ns, ns_w, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    monkeypatch.setenv('SAFE_HIP_MFMA_FILTER', '0')
    ns6, cn6, cp6, _ = _counts(amd, ctx, nbr, b, nperm, seed)
    assert be.last_mfma_filter(ctx)[0] == 6
    np.testing.assert_array_equal(ns, ns6)                            # the same exact integer sums, the same scale
    np.testing.assert_array_equal(cn, cn6)
    np.testing.assert_array_equal(cp, cp6)
    nbr.close()


def test_filtered_form_when_the_high_digits_decide_nothing(amd, ctx):
    """Columns whose values differ only in their LOW digits (a large common offset plus small noise; one column is the
    same value everywhere: every compare a tie): the high digits of every permuted sum equal the observed ones, so every
    compare goes through the exact resolve kernel -- counts still equal the oracle's."""
    from safepy_amd import backend as be
    rng = np.random.default_rng(7)
    n, m, nperm, seed = 320, 32, 40, 3
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.12)
    # 2^40 + integers below 2^22: exactly representable (41 bits: six slices), high digits (bits 24+) all equal
    b = (2.0 ** 40 + rng.integers(0, 1 << 22, size=(n, m))).astype(np.float64)
    b[:, 5] = 2.0 ** 40 + 12345.0
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.12))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    assert name == 'k_permtest_mfma' and be.last_mfma_slices(ctx) == 6
    core, undecided = be.last_mfma_filter(ctx)
    assert core == 3 and undecided > 0.5 * n * m * nperm                # (every row is present: all sums of a neighborhood share their high digits)
    np.testing.assert_array_equal(ns, orc.compute_neighborhood_score(a, b, 'sum'))
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_filtered_form_falls_back_to_six_slices_when_its_list_overflows(amd, ctx, monkeypatch):
    from safepy_amd import backend as be
    xy, a, b = _filter_case(33, n=400, m=33)
    b[:, 3] = np.round(b[:, 3] * 4) / 4 + 2.0 ** 30                    # a column with many exactly equal sums (ties are undecided)
    nperm, seed = 30, 8
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    monkeypatch.setenv('SAFE_HIP_MFMA_FILTER_CAP', '4')
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed)
    core, undecided = be.last_mfma_filter(ctx)
    assert core == 6 and undecided < 0                                  # the filtered pass was abandoned
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()


def test_filtered_form_column_shards_and_launch_spans(amd, ctx):
    """A column shard that starts inside a tile, more permutations than one launch holds."""
    xy, a, b = _filter_case(9, n=520, m=70)
    nperm, seed = 300, 21
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.1))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, col0=13, col1=61)
    assert name == 'k_permtest_mfma'
    np.testing.assert_array_equal(cn, cn_w[:, 13:61])
    np.testing.assert_array_equal(cp, cp_w[:, 13:61])
    nbr.close()


def test_filtered_form_fall_back_inside_compute_pvalues_counts_enriched_neighborhoods_once(amd, monkeypatch):
    """The abandoned filtered pass has already added its hits to the per-attribute counters when the six-slice pass starts:
    they are cleared (num_neighborhoods_enriched was doubled before)."""
    monkeypatch.setenv('SAFE_HIP_MFMA_FILTER_CAP', '4')
    monkeypatch.setenv('SAFE_HIP_FORCE_PATH', 'mfma')
    rng = np.random.default_rng(12)
    n, m, nperm = 600, 40, 50
    xy = rng.uniform(size=(n, 2))
    b = _quant(rng, n, m, np.float64, 'C', nan_rows=20, nan_frac=0.02)
    a = orc.neighborhoods_euclidean(xy, 0.1)
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=nperm, random_seed=6)
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.random_seed = 6
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
    sf.load_attributes(attribute_file=b.copy())
    sf.compute_pvalues(num_permutations=nperm, verbose=False)
    from safepy_amd import backend as be
    assert be.last_mfma_filter(amd.Context.default(0))[1] < 0
    np.testing.assert_array_equal(sf.nes_binary, want['nes_binary'])
    np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, want['nes_binary'].sum(axis=0))


# ---------------------------------------------------------------------------------------------------------------------
# z-scores, filtered: 4 of the 7 slices on the matrix cores
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('form', ['', 'gz64', 'general'])
@pytest.mark.parametrize('kind', ['normal', 'dyadic'])
def test_zscore_filtered_form_equals_the_seven_slice_form_and_the_oracle(amd, ctx, monkeypatch, kind, form):
    """form: '' = k_permtest_mfma_gz with the single-precision decision (the default), 'gz64' = the same kernel with the f64
    decision, 'general' = the general kernel's filtered form -- all three leave the seven-slice kernel's counters."""
    from safepy_amd import backend as be
    if form:
        monkeypatch.setenv('SAFE_HIP_MFMA_FORM', form)
    rng = np.random.default_rng(41)
    n, m, nperm, seed = 800, 29, 90, 2
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.09)
    b = _zdata(rng, n, m, kind, np.float64, 'C')
    cn_w, cp_w = orc.run_permutations(a, b, 'z-score', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.09))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name == 'k_permtest_mfma'
    core, undecided = be.last_mfma_filter(ctx)
    assert core == 4 and undecided >= 0
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    monkeypatch.setenv('SAFE_HIP_MFMA_FILTER', '0')
    ns7, cn7, cp7, _ = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert be.last_mfma_filter(ctx)[0] == 7
    np.testing.assert_array_equal(ns, ns7)
    np.testing.assert_array_equal(cn, cn7)
    np.testing.assert_array_equal(cp, cp7)
    nbr.close()


def test_zscore_filtered_form_with_few_distinct_values_and_tiny_neighborhoods(amd, ctx):
    """Small integers with many zeros and missing values: equal z-scores (ties count on both sides), zero variance and fewer
    than three values (NaN scores) all over -- what the filter cannot decide goes through the exact resolve kernel, or the
    call falls back to seven slices; either way the counts are the oracle's."""
    rng = np.random.default_rng(8)
    n, m, nperm, seed = 500, 24, 40, 6
    xy = rng.uniform(size=(n, 2))
    a = orc.neighborhoods_euclidean(xy, 0.05)                          # ~4 members on average: many neighborhoods below 3 values
    b = rng.integers(0, 4, size=(n, m)).astype(np.float64)
    b[rng.uniform(size=(n, m)) < 0.3] = np.nan
    b[:, 7] = 2.0                                                       # a constant column: zero variance everywhere
    ns_w = orc.compute_neighborhood_score(a, b, 'z-score')
    cn_w, cp_w = orc.run_permutations(a, b, 'z-score', nperm, seed)
    nbr = amd.Neighborhoods.euclidean(ctx, xy, orc.layout_radius(xy[:, 0], 0.05))
    ns, cn, cp, name = _counts(amd, ctx, nbr, b, nperm, seed, score='z-score')
    assert name == 'k_permtest_mfma'
    assert np.array_equal(np.isnan(ns), np.isnan(ns_w))
    np.testing.assert_array_equal(cn, cn_w)
    np.testing.assert_array_equal(cp, cp_w)
    nbr.close()
