"""Randomised form-against-form check of the quantitative permutation test (GPU only, no oracle: any size).

For random layouts, radii, data kinds, sizes, permutation counts and seeds the counters of the default kernels
(k_permtest_mfma_g / k_permtest_mfma_gz: three / four high slices on the matrix cores + the exact resolve) are compared with
  (a) the general kernel running ALL slices (SAFE_HIP_MFMA_FILTER=0, SAFE_HIP_MFMA_FORM=general), and
  (b) the f64 kernels (SAFE_HIP_FORCE_PATH=gather),
which tests/test_gpu_mfma.py and tests/test_gpu_parity.py pin to the oracle.  Equality is exact for (a); for (b) exact where the
fixed-point grid holds the data exactly (integers, dyadic values), else the f64 kernel's own sums round and single counters may
differ by the reference's own rounding (reported, not failed).

    python tools/r6/fuzz_forms.py [seconds] [first_case]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import safepy_amd                                      # noqa: E402
from safepy_amd import backend as be                   # noqa: E402


def layout(rng, n):
    kind = rng.integers(0, 3)
    if kind == 0:
        return rng.uniform(size=(n, 2))
    if kind == 1:                                       # clusters of very different density
        k = int(rng.integers(2, 12))
        centres = rng.uniform(size=(k, 2))
        spread = rng.uniform(0.005, 0.15, size=k)
        which = rng.integers(0, k, size=n)
        return centres[which] + rng.normal(size=(n, 2)) * spread[which, None]
    xy = rng.uniform(size=(n, 2))                       # a line + a blob: long thin neighborhoods
    xy[: n // 2, 1] = 0.5 + 1e-3 * rng.normal(size=n // 2)
    return xy


def data(rng, n, m, kind):
    if kind == 'normal':
        b = rng.normal(size=(n, m))
    elif kind == 'lognormal':
        b = np.exp(rng.normal(size=(n, m)) * rng.uniform(0.5, 3.0))
    elif kind == 'integers':
        b = rng.integers(-1000, 1000, size=(n, m)).astype(np.float64)
    elif kind == 'dyadic':
        b = rng.integers(-40, 41, size=(n, m)) / 8.0
    elif kind == 'few':                                 # few distinct values: ties everywhere
        b = rng.integers(0, 3, size=(n, m)).astype(np.float64) * rng.uniform(0.1, 10.0)
    elif kind == 'offset':                              # large common offset + small spread: the high digits decide little
        b = 1000.0 + rng.normal(size=(n, m)) * rng.uniform(1e-4, 1.0)
    elif kind == 'mixed':                               # columns of very different scale in one matrix
        b = rng.normal(size=(n, m)) * np.exp(rng.normal(size=m) * 6.0)[None, :]
    else:                                               # 'sparse': mostly zero
        b = rng.normal(size=(n, m)) * (rng.uniform(size=(n, m)) < 0.05)
    return np.ascontiguousarray(b)


def counts(ctx, nbr, b, nperm, seed, score, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        attr = be.Attributes.from_host(ctx, b)
        n, m = b.shape
        perms = be.Permutations(ctx, n, attr.row_flags(), nperm, seed)
        ns, neg, pos = (ctx.alloc_f64(n, m) for _ in range(3))
        be.permtest_counts(ctx, nbr, attr, perms, score, ns.ptr, neg.ptr, pos.ptr, 0, m)
        name = ctx.last_kernel()[0]
        filt = be.last_mfma_filter(ctx)
        out = (ns.download((n, m)), neg.download((n, m)), pos.download((n, m)), name, filt)
        perms.close()
        attr.close()
        return out
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run(budget, first=0, max_cases=None, narrow_lds='0'):
    """Returns (cases run, failures, cases whose counters differ from the f64 kernels' on inexact data, kernels used)."""
    ctx = safepy_amd.Context.default(0)
    os.environ['SAFE_HIP_NARROW_LDS'] = narrow_lds
    kinds = ['normal', 'lognormal', 'integers', 'dyadic', 'few', 'offset', 'mixed', 'sparse']
    t0 = time.time()
    case = first
    fails = 0
    f64_diffs = 0
    used = {}
    while time.time() - t0 < budget and (max_cases is None or case - first < max_cases):
        rng = np.random.default_rng(1000 + case)
        n = int(rng.choice([rng.integers(260, 700), rng.integers(700, 3000), rng.integers(3000, 9000)]))
        m = int(rng.choice([rng.integers(1, 8), rng.integers(8, 70), rng.integers(70, 300)]))
        nperm = int(rng.choice([rng.integers(1, 8), rng.integers(8, 120), rng.integers(120, 600)]))
        seed = int(rng.integers(0, 2 ** 32))
        score = 'z-score' if rng.uniform() < 0.5 else 'sum'
        kind = kinds[int(rng.integers(0, len(kinds)))]
        xy = layout(rng, n)
        # radius: expected neighborhood sizes from ~3 to ~1500 members (the filtered kernels need < 2048)
        diam = float(np.hypot(np.ptp(xy[:, 0]), np.ptp(xy[:, 1])))
        radius = diam * float(np.exp(rng.uniform(np.log(0.01), np.log(0.25))))
        b = data(rng, n, m, kind)
        if score == 'z-score' or rng.uniform() < 0.3:
            b[rng.uniform(size=(n, m)) < rng.choice([0.0, 0.02, 0.3])] = np.nan
            if rng.uniform() < 0.3:
                b[rng.choice(n, max(1, n // 30), replace=False)] = np.nan
        if score == 'sum':
            b = np.nan_to_num(b)                       # the reference's 'sum' path sees NaN as 0 (safe.py:478)
        nbr = safepy_amd.Neighborhoods.euclidean(ctx, xy, radius)
        tag = 'case %d: n=%d m=%d perms=%d %s %s radius=%.3g' % (case, n, m, nperm, score, kind, radius)
        try:
            d = counts(ctx, nbr, b, nperm, seed, score, {})
            g = counts(ctx, nbr, b, nperm, seed, score, {'SAFE_HIP_MFMA_FILTER': '0', 'SAFE_HIP_MFMA_FORM': 'general'})
            key = (d[3], d[4][0])
            used[key] = used.get(key, 0) + 1
            ok = np.array_equal(d[1], g[1]) and np.array_equal(d[2], g[2]) and np.array_equal(d[0], g[0], equal_nan=True)
            if not ok:
                fails += 1
                bad = np.argwhere((d[1] != g[1]) | (d[2] != g[2]))
                print('FAIL', tag, 'kernels', d[3], d[4], g[3], g[4], 'first differing outputs', bad[:5].tolist(), flush=True)
            f = counts(ctx, nbr, b, nperm, seed, score, {'SAFE_HIP_FORCE_PATH': 'gather'})
            same = np.array_equal(d[1], f[1]) and np.array_equal(d[2], f[2])
            if not same:
                nd = int(((d[1] != f[1]) | (d[2] != f[2])).sum())
                exact_kind = kind in ('integers', 'dyadic')
                if exact_kind and d[3] == 'k_permtest_mfma':
                    fails += 1
                    print('FAIL (f64 kernels, exact data)', tag, nd, 'outputs differ', flush=True)
                else:
                    f64_diffs += 1
                    print('note', tag, ': %d of %d outputs differ from the f64 kernels (%s; their sums round)' % (nd, n * m, d[3]), flush=True)
        finally:
            nbr.close()
        case += 1
    print('cases %d..%d, failures %d, cases with f64-kernel rounding differences %d, kernels used %s'
          % (first, case - 1, fails, f64_diffs, sorted(used.items())), flush=True)
    return case - first, fails, f64_diffs, used


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(budget, first)[1] else 0)


if __name__ == '__main__':
    main()
