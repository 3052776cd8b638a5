"""The narrower way in for 0/1 attribute matrices (additive to safe_io.py:361 / safe.py:556-608): a uint8 or bool
`node2attribute` travels as bytes (SAFE_DTYPE_U8) and gives exactly the results of the same matrix as f32 / f64 --
hypergeometric and randomization paths, C and Fortran order -- and the oracle's.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


def _instance(amd, n, seed):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(size=(n, 2))
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy, np.zeros(0, np.int32), np.zeros(0, np.int32))
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.12)
    return sf, rng


OUTS = ('pvalues_pos', 'nes', 'nes_binary')


@pytest.mark.parametrize('order', ['C', 'F'])
@pytest.mark.parametrize('dtype', [np.uint8, np.bool_])
def test_u8_matrix_equals_f32_hypergeometric_and_randomization(amd, dtype, order):
    n, m = 700, 333
    sf, rng = _instance(amd, n, 11)
    b01 = (rng.uniform(size=(n, m)) < 0.04)
    want = {}
    for how, kw in (('hypergeometric', {}), ('randomization', {'num_permutations': 120})):
        sf.random_seed = 3
        sf.load_attributes(attribute_file=np.asarray(b01.astype(np.float32), order=order))
        sf.compute_pvalues(how=how, **kw)
        want[how] = {k: getattr(sf, k).copy() for k in OUTS}
        want[how]['enriched'] = sf.attributes['num_neighborhoods_enriched'].values.copy()
    narrow = np.asarray(b01.astype(dtype), order=order)
    for how, kw in (('hypergeometric', {}), ('randomization', {'num_permutations': 120})):
        sf.random_seed = 3
        sf.load_attributes(attribute_file=narrow)
        sf.compute_pvalues(how=how, **kw)
        for k in OUTS:
            assert np.array_equal(getattr(sf, k), want[how][k], equal_nan=True), (how, k)
        assert np.array_equal(sf.attributes['num_neighborhoods_enriched'].values, want[how]['enriched']), how
    # 'auto' sees a binary matrix (safe.py:463) and the network background leaves a matrix without missing values alone
    sf.load_attributes(attribute_file=narrow)
    sf.compute_pvalues(how='auto', background='network')
    assert sf.enrichment_type == 'auto' and np.array_equal(sf.nes, want['hypergeometric']['nes'], equal_nan=True)
    assert narrow.dtype == dtype and np.array_equal(narrow.astype(bool), b01)          # the caller's matrix is untouched


def test_u8_matrix_against_the_oracle(amd):
    n, m = 500, 97
    sf, rng = _instance(amd, n, 5)
    b = (rng.uniform(size=(n, m)) < 0.06).astype(np.uint8)
    sf.load_attributes(attribute_file=b)
    sf.compute_pvalues(how='hypergeometric')
    ref = orc.pvalues_by_hypergeom(np.asarray(sf.neighborhoods), b.astype(np.float64))
    np.testing.assert_allclose(sf.pvalues_pos, ref['pvalues_pos'], rtol=1e-6, atol=1e-300)     # tolerance of north_star (1e-6 relative)
    np.testing.assert_allclose(sf.nes, ref['nes'], rtol=1e-6, atol=1e-9)


def test_u8_upload_is_a_host_form_only(amd):
    from safepy_amd import backend as be, _lib
    import ctypes as C
    ctx = be.Context.default(0)
    h = C.c_void_p()
    rc = _lib.lib.safe_attr_create_dev(ctx.handle, C.c_void_p(16), _lib.DTYPE_U8, 4, 4, 4, 1, C.byref(h))
    assert rc != 0 and b'f32 or f64' in _lib.lib.safe_last_error()
