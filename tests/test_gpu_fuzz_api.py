"""Random whole-call cases of SAFE.compute_pvalues() against the oracle (oracle/safe_oracle.py, which follows safe.py:432-554 and
safe_extras.py:6-70): memberships from random layouts or random dense matrices (asymmetric, empty rows, isolated nodes), binary /
small-integer / dyadic / continuous attributes in f32 / f64, C / F order, NaN cells and NaN rows, every enrichment type, score
type, sign, background and multiple_testing setting, odd permutation counts.  Counts are compared EXACTLY, hypergeometric
p-values to 1e-6 relative.  The one licence: on continuous data a permutation that puts the SAME multiset of three or more values
into a neighborhood (tiny networks: 3 members of 20 nodes, 257 permutations) ties mathematically, and then the reference's own
`<=` / `>=` is decided by the order its BLAS adds them -- a counter may differ from the oracle's by at most the
number of such permutations of that neighborhood, which the test counts from the oracle's own index table.

SAFE_FUZZ_SECONDS (default 40) bounds the run; SAFE_FUZZ_FIRST names the first case (cases are seeded by their number)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


def _case(rng):
    n = int(rng.choice([rng.integers(3, 40), rng.integers(40, 300), rng.integers(300, 900)]))
    m = int(rng.choice([1, rng.integers(2, 10), rng.integers(10, 80)]))
    if rng.uniform() < 0.5:
        xy = rng.uniform(size=(n, 2))
        a = orc.neighborhoods_euclidean(xy, float(rng.choice([0.02, 0.08, 0.15, 0.4])))
    else:
        a = (rng.uniform(size=(n, n)) < rng.choice([0.01, 0.05, 0.3])).astype(np.int64)
        if n > 5:
            a[int(rng.integers(0, n)), :] = 0                      # an empty neighborhood
            a[:, int(rng.integers(0, n))] = 0                      # a node nobody contains
    kind = str(rng.choice(['binary', 'binary', 'integers', 'dyadic', 'normal']))
    if kind == 'binary':
        b = (rng.uniform(size=(n, m)) < np.exp(rng.uniform(np.log(0.01), np.log(0.8), size=m))[None, :]).astype(np.float64)
    elif kind == 'integers':
        b = rng.integers(-5, 6, size=(n, m)).astype(np.float64)
    elif kind == 'dyadic':
        b = rng.integers(-40, 41, size=(n, m)) / 8.0
    else:
        b = rng.normal(size=(n, m))
    if rng.uniform() < 0.6:
        b[rng.uniform(size=(n, m)) < rng.choice([0.01, 0.2])] = np.nan
    if rng.uniform() < 0.4 and n > 4:
        b[rng.choice(n, max(1, n // 10), replace=False)] = np.nan
    if m > 2 and rng.uniform() < 0.5:
        b[:, 0] = np.nan                                           # nothing annotated at all
        b[:, 1] = 0 if kind == 'binary' else 2.0                   # a constant column
    if rng.uniform() < 0.4:
        b = b.astype(np.float32)
    if rng.uniform() < 0.5:
        b = np.asfortranarray(b)
    kw = dict(how=str(rng.choice(['auto', 'randomization', 'randomization', 'hypergeometric'])) if kind == 'binary'
              else str(rng.choice(['auto', 'randomization'])),
              neighborhood_score_type=str(rng.choice(['sum', 'z-score'])),
              background=str(rng.choice(['attribute_file', 'network'])),
              multiple_testing=bool(rng.uniform() < 0.3),
              num_permutations=int(rng.choice([10, 11, 33, 100, 257])))       # (the reference refuses fewer than 10: safe.py:211)
    sign = str(rng.choice(['both', 'highest', 'lowest']))
    seed = int(rng.integers(0, 2 ** 32))
    thr = float(rng.choice([0.05, 0.2]))
    return a, b, kind, kw, sign, seed, thr


def _tied_permutations(a, b, kw, seed):
    """[N, M]: for every output the number of permutations that put the observed multiset of values into the neighborhood."""
    work = np.array(b, dtype=np.float64, order='C')
    if kw['background'] == 'network':
        work[np.isnan(work)] = 0                                    # safe.py:449-451
    table = orc.permutation_index_table(work, kw['num_permutations'], seed)
    n, m = work.shape
    ties = np.zeros((n, m), dtype=np.int64)
    members = [np.flatnonzero(a[i]) for i in range(n)]
    for i in range(n):
        if len(members[i]) < 3:
            continue                                                # sums of one or two values do not depend on the order
        obs = np.sort(work[members[i]], axis=0)
        for k in range(table.shape[0]):
            per = np.sort(work[table[k][members[i]]], axis=0)
            ties[i] += np.all((obs == per) | (np.isnan(obs) & np.isnan(per)), axis=0)
    return ties


def _counts_equal_up_to_ties(got, want, nperm, a, b, kw, seed, kind, tag):
    if np.array_equal(got, want, equal_nan=True):
        return
    assert np.array_equal(np.isnan(got), np.isnan(want)), tag
    ties = _tied_permutations(a, b, kw, seed)
    diff = np.abs(np.nan_to_num(got) - np.nan_to_num(want)) * nperm
    bad = np.argwhere(diff > ties + 1e-6)
    assert bad.size == 0, '%s: counters differ beyond the mathematically tied permutations at %s' % (tag, bad[:5].tolist())


def test_random_whole_calls_against_the_oracle():
    import safepy_amd as amd
    assert amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    budget = float(os.environ.get('SAFE_FUZZ_SECONDS', '40'))
    first = int(os.environ.get('SAFE_FUZZ_FIRST', '0'))
    t0, case, seen, licensed = time.time(), first, set(), 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(900000 + case)
        a, b, kind, kw, sign, seed, thr = _case(rng)
        tag = 'case %d: n=%d m=%d %s %s sign=%s seed=%d thr=%g %s' % (case, b.shape[0], b.shape[1], kind, kw, sign, seed, thr, b.dtype)
        want = orc.compute_pvalues(a, np.array(b, order='C'), enrichment_type=kw['how'],       # (same dtype: np.power keeps f32, safe_extras.py:24)
                                   neighborhood_score_type=kw['neighborhood_score_type'], background=kw['background'],
                                   num_permutations=kw['num_permutations'], random_seed=seed, attribute_sign=sign,
                                   enrichment_threshold=thr, multiple_testing=kw['multiple_testing'])
        sf = amd.SAFE(verbose=False)
        sf.attribute_sign = sign
        sf.random_seed = seed
        sf.enrichment_threshold = thr
        sf.neighborhoods = a
        sf.load_attributes(attribute_file=b.copy(order='K'))
        sf.compute_pvalues(verbose=False, **kw)
        hyper = 'ns' not in want
        seen.add((kind, 'hyper' if hyper else kw['neighborhood_score_type']))
        if hyper:
            assert sf.ns is None and sf.pvalues_neg is None, tag
            np.testing.assert_allclose(sf.pvalues_pos, want['pvalues_pos'], rtol=1e-6, atol=1e-300, err_msg=tag)
            np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-6, atol=1e-9, err_msg=tag)
        else:
            if kind == 'normal':
                np.testing.assert_allclose(sf.ns, want['ns'], rtol=1e-6, atol=1e-9, equal_nan=True, err_msg=tag)
            else:
                np.testing.assert_array_equal(sf.ns, want['ns'], err_msg=tag)
            tol = dict(rtol=1e-12, atol=0) if kw['multiple_testing'] else dict(rtol=0, atol=0)
            same = all(np.allclose(getattr(sf, key), want[key], equal_nan=True, **tol) for key in ('pvalues_neg', 'pvalues_pos'))
            if not same:
                assert kind == 'normal', tag                        # exact data: no licence
                licensed += 1
                if kw['multiple_testing']:                          # the licence is about the counters: the same call without the correction
                    kw = dict(kw, multiple_testing=False)
                    want = orc.compute_pvalues(a, np.array(b, order='C'), enrichment_type=kw['how'],
                                               neighborhood_score_type=kw['neighborhood_score_type'], background=kw['background'],
                                               num_permutations=kw['num_permutations'], random_seed=seed, attribute_sign=sign,
                                               enrichment_threshold=thr, multiple_testing=False)
                    sf.compute_pvalues(verbose=False, **kw)
                _counts_equal_up_to_ties(sf.pvalues_neg, want['pvalues_neg'], kw['num_permutations'], a, b, kw, seed, kind, tag)
                _counts_equal_up_to_ties(sf.pvalues_pos, want['pvalues_pos'], kw['num_permutations'], a, b, kw, seed, kind, tag)
                case += 1
                continue                                            # (nes / nes_binary follow the counters)
            if kw['multiple_testing']:
                np.testing.assert_allclose(sf.nes, want['nes'], rtol=1e-9, atol=1e-12, err_msg=tag)
            else:
                np.testing.assert_array_equal(sf.nes, want['nes'], err_msg=tag)
        with np.errstate(invalid='ignore'):
            clear = ~(np.abs(np.abs(want['nes']) - (-np.log10(thr))) < 1e-6)       # (NaN compares False: those cells are checked too)
        np.testing.assert_array_equal(sf.nes_binary[clear], want['nes_binary'][clear], err_msg=tag)
        if clear.all():
            np.testing.assert_array_equal(sf.attributes['num_neighborhoods_enriched'].values, want['num_neighborhoods_enriched'], err_msg=tag)
        case += 1
    print('cases %d..%d, %d of them with counters inside the tie licence' % (first, case - 1, licensed))
    assert case - first >= 20, 'only %d cases in %g s' % (case - first, budget)
    assert licensed * 20 <= case - first, 'the tie licence was needed in %d of %d cases' % (licensed, case - first)
    assert len(seen) >= 6, seen
