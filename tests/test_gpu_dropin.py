"""The drop-in class as a notebook uses it: lazy result attributes and their recycled host arrays (safepy_amd/safe.py _LazyArray).
Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


def _instance(amd, seed):
    rng = np.random.default_rng(seed)
    n, m = 400, 37
    xy = rng.uniform(size=(n, 2))
    b = (rng.uniform(size=(n, m)) < 0.06).astype(np.float64)
    b[rng.choice(n, 11, replace=False)] = np.nan
    sf = amd.SAFE(verbose=False)
    sf.graph = amd.LayoutGraph(xy)
    sf.random_seed = 3
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
    sf.load_attributes(attribute_file=b.copy())
    return sf, orc.neighborhoods_euclidean(xy, 0.1), b


def test_result_arrays_the_caller_keeps_are_never_written_again(amd):
    sf, a, b = _instance(amd, 1)
    sf.compute_pvalues(how='randomization', num_permutations=30, verbose=False)
    want30 = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=30, random_seed=3)
    kept = sf.nes                                              # the caller holds on to the first call's result
    kept_copy = kept.copy()
    np.testing.assert_array_equal(kept, want30['nes'])
    sf.compute_pvalues(how='randomization', num_permutations=60, verbose=False)
    want60 = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=60, random_seed=3)
    second = sf.nes
    assert second is not kept                                  # a new array, as in the reference (safe.py:549-554)
    np.testing.assert_array_equal(second, want60['nes'])
    np.testing.assert_array_equal(kept, kept_copy)             # ... and the kept one is untouched


def test_result_arrays_nobody_holds_are_recycled_with_the_right_values(amd):
    sf, a, b = _instance(amd, 2)
    sf.compute_pvalues(how='randomization', num_permutations=30, verbose=False)
    first_id = id(sf.nes)                                      # read (copied to the host), no reference kept
    first_bin = id(sf.nes_binary)
    sf.compute_pvalues(how='randomization', num_permutations=50, verbose=False)
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=50, random_seed=3)
    assert id(sf.nes) == first_id and id(sf.nes_binary) == first_bin       # the same host memory, written again
    np.testing.assert_array_equal(sf.nes, want['nes'])
    np.testing.assert_array_equal(sf.nes_binary, want['nes_binary'])
    np.testing.assert_array_equal(sf.pvalues_pos, want['pvalues_pos'])
    # the caller clears the attribute (bench.py's pattern) and computes again with another shape of work
    sf.nes = None
    sf.compute_pvalues(how='hypergeometric')
    wanth = orc.compute_pvalues(a, b.copy(), enrichment_type='hypergeometric')
    assert id(sf.nes) == first_id
    np.testing.assert_allclose(sf.nes, wanth['nes'], rtol=1e-6, atol=1e-12)
    np.testing.assert_array_equal(sf.nes_binary, wanth['nes_binary'])


def test_a_value_the_caller_assigned_is_not_recycled(amd):
    sf, a, b = _instance(amd, 3)
    sf.compute_pvalues(how='randomization', num_permutations=20, verbose=False)
    mine = np.full(sf.nes.shape, 7.0)
    sf.nes = mine                                              # the caller's own array in the attribute
    sf.compute_pvalues(how='randomization', num_permutations=20, verbose=False)
    assert sf.nes is not mine
    assert (mine == 7.0).all()


@pytest.mark.parametrize('threads', ['0', '1', '4', '16'])
def test_threaded_read_back_of_a_large_buffer_with_an_odd_tail(amd, monkeypatch, threads):
    """safe_memcpy_d2h takes its pinned ring + copy threads from 64 MB on: a buffer of 64 MB + an odd tail (a last slot of
    a few KB, not a multiple of anything) must arrive byte for byte with 1, 4 and 16 copy threads, as with the plain copy
    (SAFE_HIP_D2H_THREADS=0) and the resident-pages copy."""
    import ctypes as C
    from safepy_amd import backend as be
    ctx = amd.Context.default(0)
    n_words = (64 << 20) // 8 + 12345
    src = (np.arange(n_words, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.float64)     # every word distinct
    buf = ctx.alloc_f64(n_words)
    be.check(be.lib.safe_memcpy_h2d(ctx.handle, C.c_void_p(buf.ptr), src.ctypes.data_as(C.c_void_p), C.c_size_t(src.nbytes)))
    monkeypatch.setenv('SAFE_HIP_D2H_THREADS', threads)
    got = buf.download((n_words,))
    assert np.array_equal(got.view(np.uint64), src.view(np.uint64))
    again = buf.download((n_words,), out=got)                  # the plain copy into resident pages
    assert again is got and np.array_equal(got.view(np.uint64), src.view(np.uint64))
    buf.free()
