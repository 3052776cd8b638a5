"""Randomised form-against-form check of the BINARY permutation test (GPU only, no oracle: any size).

Binary 'sum' scores are exact in integers, so every kernel family must leave the same counters for the same seed:
  default (k_permtest_bits_blk where its limits hold, else what choose_path picks), the f64 gather kernel, the sparse scatter
  kernel, the pre-permuted, the LDS-row and the stream-less bit-sliced kernels (a forced form that does not apply to a shape
  falls through to what choose_path picks; the summary line says which kernel ran how often).
The kernels themselves are pinned to the oracle by tests/test_gpu_parity.py; this tool walks shapes those tests do not: random
sizes up to 32 767 nodes (past the blocked kernel's N <= 8190: the sixteen-wave forms, full and half words), clustered layouts, dense random memberships with rows of every
size class, NaN rows, empty and full columns, odd permutation counts.

    python tools/r6/fuzz_binary.py [seconds] [first_case]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import safepy_amd                                      # noqa: E402
from safepy_amd import backend as be                   # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_forms import layout, counts                  # noqa: E402

FORMS = [('gather', {'SAFE_HIP_FORCE_PATH': 'gather'}),
         ('scatter', {'SAFE_HIP_FORCE_PATH': 'scatter'}),
         ('bits-pre', {'SAFE_HIP_FORCE_PATH': 'bits', 'SAFE_HIP_BITS_KERNEL': 'pre'}),
         ('bits-row', {'SAFE_HIP_FORCE_PATH': 'bits', 'SAFE_HIP_BITS_PRE': '0'}),
         ('bits-plain', {'SAFE_HIP_FORCE_PATH': 'bits', 'SAFE_HIP_BITS_DBG': '256'})]


def run(budget, first=0, max_cases=None):
    """Returns (cases run, failures, {(form, kernel name): cases})."""
    ctx = safepy_amd.Context.default(0)
    os.environ['SAFE_HIP_NARROW_LDS'] = '0'
    t0 = time.time()
    case = first
    fails = 0
    used = {}
    while time.time() - t0 < budget and (max_cases is None or case - first < max_cases):
        rng = np.random.default_rng(500000 + case)
        dense = rng.uniform() < 0.35
        if dense:
            n = int(rng.integers(40, 2500))
        else:
            n = int(rng.choice([rng.integers(20, 300), rng.integers(300, 3000), rng.integers(3000, 8191), rng.integers(8191, 32768)]))
        m = int(rng.choice([rng.integers(1, 8), rng.integers(8, 130), rng.integers(130, 700)]))
        nperm = int(rng.choice([rng.integers(1, 8), rng.integers(8, 120), rng.integers(120, 700)]))
        seed = int(rng.integers(0, 2 ** 32))
        if dense:                                       # rows of every width class of the bit-sliced kernels, asymmetric
            a = np.zeros((n, n), dtype=np.int64)
            top = int(rng.choice([8, 56, 248, 504, 1000, 1500, 2047, 2300]))
            sizes = np.minimum(rng.integers(0, top + 1, size=n), n)
            for i, k in enumerate(sizes):
                a[i, rng.choice(n, int(k), replace=False)] = 1
            nbr = safepy_amd.Neighborhoods.from_dense(ctx, a)
            what = 'dense rows <= %d' % top
        else:
            xy = layout(rng, n)
            diam = float(np.hypot(np.ptp(xy[:, 0]), np.ptp(xy[:, 1])))
            radius = diam * float(np.exp(rng.uniform(np.log(0.005), np.log(0.2))))
            nbr = safepy_amd.Neighborhoods.euclidean(ctx, xy, radius)
            what = 'radius %.3g' % radius
        dens = np.exp(rng.uniform(np.log(0.002), np.log(0.9), size=m))
        b = (rng.uniform(size=(n, m)) < dens[None, :]).astype(np.float32 if rng.uniform() < 0.5 else np.float64)
        if m > 3:
            b[:, 1] = 0
            b[:, 2] = 1
        if rng.uniform() < 0.5:
            b[rng.choice(n, max(1, n // int(rng.integers(5, 60))), replace=False)] = np.nan       # rows the permutations leave in place
        if rng.uniform() < 0.5:
            b = np.asfortranarray(b)
        tag = 'case %d: n=%d m=%d perms=%d %s' % (case, n, m, nperm, what)
        try:
            d = counts(ctx, nbr, b, nperm, seed, 'sum', {})
            used[('default', d[3])] = used.get(('default', d[3]), 0) + 1
            for form, env in FORMS:
                f = counts(ctx, nbr, b, nperm, seed, 'sum', env)
                used[(form, f[3])] = used.get((form, f[3]), 0) + 1
                if not (np.array_equal(d[1], f[1]) and np.array_equal(d[2], f[2]) and np.array_equal(d[0], f[0], equal_nan=True)):
                    fails += 1
                    bad = np.argwhere((d[1] != f[1]) | (d[2] != f[2]))
                    print('FAIL', tag, ': default', d[3], 'against', form, f[3], 'first differing outputs', bad[:5].tolist(), flush=True)
        finally:
            nbr.close()
        case += 1
    print('cases %d..%d, failures %d, forms %s' % (first, case - 1, fails, sorted(used.items())), flush=True)
    return case - first, fails, used


if __name__ == '__main__':
    sys.exit(1 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)[1] else 0)
