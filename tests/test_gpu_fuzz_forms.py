"""Short runs of tools/r6/fuzz_forms.py and tools/r6/fuzz_binary.py.  fuzz_forms: random layouts, radii, data kinds, sizes, permutation counts and seeds; the counters
of the default matrix-core kernels (filtered: high slices + exact resolve) must equal the general kernel's with ALL slices, and
the f64 kernels' wherever the data is exact on both grids.  fuzz_binary: 0/1 attributes, every kernel family (blocked / pre-permuted /
LDS-row / stream-less bit-sliced, scatter, f64 gather) against the default.  (The long runs -- 4000 and 490 cases, 6-7 minutes each,
0 failures at the end of round 6 -- are the tools themselves.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    import sys
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'r6'))
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', 'r6', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_cases_every_form_leaves_the_same_counters(monkeypatch):
    fuzz = _tool('fuzz_forms')
    monkeypatch.setenv('SAFE_HIP_NARROW_LDS', '0')          # run() sets it again; monkeypatch restores the caller's value afterwards
    cases, fails, _, used = fuzz.run(budget=120.0, first=20000, max_cases=250)
    assert cases >= 50 and fails == 0
    names = {k[0] for k in used}
    cores = {k[1] for k in used if k[0] == 'k_permtest_mfma'}
    assert 'k_permtest_mfma' in names and {3, 4} <= cores, used     # both filtered kernels were among the forms exercised


def test_random_binary_cases_every_kernel_family_leaves_the_same_counters(monkeypatch):
    fuzz = _tool('fuzz_binary')
    monkeypatch.setenv('SAFE_HIP_NARROW_LDS', '0')
    cases, fails, used = fuzz.run(budget=60.0, first=70000, max_cases=40)
    assert cases >= 20 and fails == 0
    ran = {k[1] for k in used}
    assert {'k_permtest_bits_blk', 'k_permtest_bits_pre', 'k_permtest_bits', 'k_permtest_scatter'} <= ran, used


@pytest.mark.parametrize('n,nperm', [(9000, 1100), (21000, 1030), (2600, 2000)])
def test_more_than_1023_permutations_on_the_sixteen_wave_forms(n, nperm, monkeypatch):
    """A task of the bit-sliced kernels counts at most 255 permutations in eight counter levels, whatever the call's count; the
    sixteen-wave forms of round 6 (full words N = 9000, half words N = 21 000, eleven sum levels for a 1300-member hub at
    N = 2600) against the f64 kernel over more than 1023 permutations (the exchange's narrow form ends there too)."""
    import numpy as np
    fuzz = _tool('fuzz_forms')
    import safepy_amd
    ctx = safepy_amd.Context.default(0)
    rng = np.random.default_rng(n)
    xy = rng.uniform(size=(n, 2))
    if n == 2600:
        xy[:1300] = xy[0] + 0.003 * rng.normal(size=(1300, 2))
    nbr = safepy_amd.Neighborhoods.euclidean(ctx, xy, 0.02)
    b = (rng.uniform(size=(n, 70)) < np.linspace(0.005, 0.6, 70)).astype(np.float32)
    b[rng.choice(n, 40, replace=False)] = np.nan
    d = fuzz.counts(ctx, nbr, b, nperm, 77, 'sum', {})
    f = fuzz.counts(ctx, nbr, b, nperm, 77, 'sum', {'SAFE_HIP_FORCE_PATH': 'gather'})
    assert d[3] == 'k_permtest_bits_pre' and f[3].startswith('k_permtest_gather')
    assert np.array_equal(d[1], f[1]) and np.array_equal(d[2], f[2]) and np.array_equal(d[0], f[0], equal_nan=True)
    assert d[1].max() == nperm
    nbr.close()
