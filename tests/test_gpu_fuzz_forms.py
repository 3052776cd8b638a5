"""A short run of tools/r6/fuzz_forms.py: random layouts, radii, data kinds, sizes, permutation counts and seeds; the counters
of the default matrix-core kernels (filtered: high slices + exact resolve) must equal the general kernel's with ALL slices, and
the f64 kernels' wherever the data is exact on both grids.  (The long run -- 4000 cases, 7 minutes -- is the tool itself.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_cases_every_form_leaves_the_same_counters(monkeypatch):
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    spec = importlib.util.spec_from_file_location('fuzz_forms', os.path.join(ROOT, 'tools', 'r6', 'fuzz_forms.py'))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setenv('SAFE_HIP_NARROW_LDS', '0')          # run() sets it again; monkeypatch restores the caller's value afterwards
    cases, fails, _, used = fuzz.run(budget=120.0, first=20000, max_cases=250)
    assert cases >= 50 and fails == 0
    names = {k[0] for k in used}
    cores = {k[1] for k in used if k[0] == 'k_permtest_mfma'}
    assert 'k_permtest_mfma' in names and {3, 4} <= cores, used     # both filtered kernels were among the forms exercised
