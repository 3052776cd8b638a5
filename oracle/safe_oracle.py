"""CPU oracle for the SAFE hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy restatement of the reference algorithm for
``define_neighborhoods`` / ``compute_pvalues`` (baryshnikova-lab/safepy).  It is
the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under ``safepy_amd/``
imports it, and the product path has no CPU fallback.

Parity status: PINNED, every branch (``multiple_testing=True`` included, see
``fdrcorrection``).  ``tests/golden/make_golden.py`` imports the real
reference from ``/root/reference`` (in the build container only), runs it on
seeded synthetic inputs and commits the input/output vectors under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors (masks/counts exact, scores <= 1e-12 relative).

Each function cites the reference file:line it follows.  Arithmetic that the
reference delegates to third-party libraries is delegated to the same library
call here (scipy ``pdist``/``squareform``, ``np.dot``, ``scipy.stats.hypergeom.sf``,
legacy ``np.random.seed`` + ``np.random.permutation``) so that the oracle *is*
the reference arithmetic; the bounded Dijkstra (networkx 3.4.2 in the reference)
is restated with ``heapq`` because networkx is not guaranteed on the GPU host.
"""
import heapq

import numpy as np
from scipy.spatial.distance import pdist, squareform
from scipy.stats import hypergeom


# ---------------------------------------------------------------------------
# Neighborhoods (reference: safepy/safe.py:369-430)
# ---------------------------------------------------------------------------

def layout_radius(x, neighborhood_radius):
    """nr = radius * (max x - min x): x-range only (safe.py:390-391, 404-405)."""
    x = np.asarray(x, dtype=np.float64)
    return neighborhood_radius * (np.max(x) - np.min(x))


def euclidean_distances(xy):
    """Dense symmetric distance matrix (safe.py:397; scipy pdist + squareform)."""
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    return squareform(pdist(xy, 'euclidean'))


def neighborhoods_euclidean(xy, neighborhood_radius):
    """A[i,j] = 1 iff D[i,j] < nr, int64, diagonal kept (safe.py:387-399, 419-420)."""
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    n = xy.shape[0]
    nr = layout_radius(xy[:, 0], neighborhood_radius)
    out = np.zeros((n, n), dtype=np.int64)
    out[euclidean_distances(xy) < nr] = 1
    return out


def _adjacency_lists(n, edge_u, edge_v, edge_w):
    adj = [[] for _ in range(n)]
    for u, v, w in zip(edge_u, edge_v, edge_w):
        u = int(u)
        v = int(v)
        adj[u].append((v, w))
        if u != v:
            adj[v].append((u, w))
    return adj


def bounded_dijkstra(adj, source, cutoff):
    """Single-source Dijkstra with cutoff, networkx semantics
    (networkx 3.4.2 algorithms/shortest_paths/weighted.py:784-880 as called from
    safe.py:406-410): a candidate distance is dropped iff it is > cutoff, so
    targets at exactly ``cutoff`` are kept; dist(source) = 0."""
    dist = {}
    seen = {source: 0}
    heap = [(0, 0, source)]
    tick = 1
    while heap:
        d, _, v = heapq.heappop(heap)
        if v in dist:
            continue
        dist[v] = d
        for u, w in adj[v]:
            cand = d + w
            if cand > cutoff:
                continue
            if u in dist:
                continue
            if u not in seen or cand < seen[u]:
                seen[u] = cand
                heapq.heappush(heap, (cand, tick, u))
                tick += 1
    return dist


def neighborhoods_shortpath(n, edge_u, edge_v, edge_w, cutoff):
    """All-pairs bounded shortest paths -> (A int64 [N,N], D f64 [N,N] with inf
    where unreached) (safe.py:403-417).  ``edge_w`` = edge 'length' for
    'shortpath_weighted_layout' (cutoff = radius * x-range), or all ones for
    'shortpath' (cutoff = radius)."""
    adj = _adjacency_lists(n, edge_u, edge_v, edge_w)
    out = np.zeros((n, n), dtype=np.int64)
    dmat = np.full((n, n), np.inf, dtype=np.float64)
    for s in range(n):
        for t, d in bounded_dijkstra(adj, s, cutoff).items():
            out[s, t] = 1
            dmat[s, t] = d
    return out, dmat


def edge_lengths(xy, edge_u, edge_v):
    """Euclidean length of every edge (safe_io.py:311-333 picks entries of the
    pdist matrix; entry-wise that is sqrt(dx*dx + dy*dy) with each op rounded)."""
    d = euclidean_distances(xy)
    return d[np.asarray(edge_u, dtype=np.int64), np.asarray(edge_v, dtype=np.int64)]


# ---------------------------------------------------------------------------
# Neighborhood score (reference: safepy/safe_extras.py:6-33)
# ---------------------------------------------------------------------------

def compute_neighborhood_score(neighborhood2node, node2attribute, neighborhood_score_type):
    with np.errstate(invalid='ignore', divide='ignore'):
        notnan = ~np.isnan(node2attribute)
        b0 = np.where(notnan, node2attribute, 0)        # safe_extras.py:10
        score = np.dot(neighborhood2node, b0)           # safe_extras.py:15
        if neighborhood_score_type == 'z-score':        # safe_extras.py:19-31
            cnt = np.dot(neighborhood2node, np.where(notnan, 1, 0))
            mean = np.divide(score, cnt)
            exx = np.divide(np.dot(neighborhood2node, np.power(b0, 2)), cnt)
            std = np.sqrt(exx - np.power(mean, 2))
            score = np.divide(mean, std)
            score[std == 0] = np.nan
            score[cnt < 3] = np.nan
    return score


# ---------------------------------------------------------------------------
# Permutation test (reference: safepy/safe_extras.py:36-70)
# ---------------------------------------------------------------------------

def run_permutations(neighborhood2node, node2attribute, neighborhood_score_type,
                     num_permutations, random_seed):
    """Returns (counts_neg, counts_pos) as float64 [N,M].  Reseeds the global
    legacy RNG (safe_extras.py:46); rows are permuted cumulatively in place and
    only rows with >= 1 non-NaN value move (safe_extras.py:50-58)."""
    np.random.seed(random_seed)
    observed = compute_neighborhood_score(neighborhood2node, node2attribute, neighborhood_score_type)
    work = np.copy(node2attribute)
    movable = np.nonzero(np.sum(~np.isnan(work), axis=1))[0]
    counts_neg = np.zeros(observed.shape)
    counts_pos = np.zeros(observed.shape)
    for _ in range(int(num_permutations)):
        work[movable, :] = work[np.random.permutation(movable), :]
        perm_score = compute_neighborhood_score(neighborhood2node, work, neighborhood_score_type)
        with np.errstate(invalid='ignore', divide='ignore'):
            counts_neg = counts_neg + (perm_score <= observed)   # safe_extras.py:65
            counts_pos = counts_pos + (perm_score >= observed)   # safe_extras.py:66
    return counts_neg, counts_pos


def permutation_index_table(node2attribute, num_permutations, random_seed):
    """Composed row-index table [P,N]: permuted matrix at iteration k equals
    node2attribute[table[k]] (follows safe_extras.py:46-58).
    Consumes the legacy global RNG exactly as run_permutations does."""
    np.random.seed(random_seed)
    n = node2attribute.shape[0]
    movable = np.nonzero(np.sum(~np.isnan(node2attribute), axis=1))[0]
    cur = np.arange(n, dtype=np.int64)
    table = np.empty((int(num_permutations), n), dtype=np.int64)
    for k in range(int(num_permutations)):
        cur[movable] = cur[np.random.permutation(movable)]
        table[k] = cur
    return table


# ---------------------------------------------------------------------------
# compute_pvalues and its two branches (reference: safepy/safe.py:432-608)
# ---------------------------------------------------------------------------

def wants_hypergeometric(node2attribute, enrichment_type):
    """Dispatch rule of safe.py:461-463."""
    other = np.sum(~np.isnan(node2attribute) & ~np.isin(node2attribute, [0, 1]))
    return (enrichment_type == 'hypergeometric') or (enrichment_type == 'auto' and other == 0)


def fdrcorrection(pvals):
    """Benjamini-Hochberg adjusted p-values of one row: what the reference gets from
    ``statsmodels.stats.multitest.fdrcorrection(pvals)[1]`` (alpha=0.05, method='indep',
    is_sorted=False), called row by row at safe.py:536-542 and 599-605.

    PINNED: ``tests/golden/fdr.npz`` holds outputs of the real ``fdrcorrection`` (statsmodels 0.12.2,
    the copy under /opt/conda in the build image, reached by appending that directory to sys.path;
    the reference pins 0.14.4 in extras/requirements.txt -- version skew, stated, the function's
    arithmetic is the same seven NumPy calls in both) on rows of length 1 ... 4373 with ties,
    zeros, ones and NaN, and of the UNSTUBBED reference run with ``multiple_testing=True``
    (randomization x {sum, z-score} x three attribute signs, hypergeometric with a NaN column,
    1 / 2 / 129 attributes per row); ``tests/test_oracle_golden.py`` checks this function and
    ``compute_pvalues(..., multiple_testing=True)`` against them bit for bit.  The restatement
    follows the published algorithm operation by operation -- argsort; ecdf = arange(1, n+1) /
    float(n); sorted / ecdf; np.minimum.accumulate from the right; clip at 1; scatter back.
    NaN p-values sort last and np.minimum propagates them through the whole accumulated row,
    exactly as the NumPy calls above would."""
    pvals = np.asarray(pvals, dtype=np.float64)
    n = pvals.shape[0]
    order = np.argsort(pvals)
    ps = np.take(pvals, order)
    ecdf = np.arange(1, n + 1) / float(n)
    with np.errstate(invalid='ignore'):
        raw = ps / ecdf
        corrected = np.minimum.accumulate(raw[::-1])[::-1]
        corrected[corrected > 1] = 1
    out = np.empty_like(corrected)
    out[order] = corrected
    return out


def fdr_rows(pvalues):
    """np.apply_along_axis(fdrcorrection, 1, pvalues)[:, 1, :] (safe.py:538-542, 604-605)."""
    return np.stack([fdrcorrection(row) for row in pvalues]) if pvalues.shape[0] else pvalues.copy()


def pvalues_by_hypergeom(neighborhoods, node2attribute, multiple_testing=False):
    """safe.py:573-608.  Returns dict(pvalues_pos, nes)."""
    n_nodes, n_attr = node2attribute.shape
    nodes_not_nan = np.any(~np.isnan(node2attribute), axis=1)
    total = np.sum(nodes_not_nan)
    pop = np.zeros([n_nodes, n_attr]) + total
    in_group = np.tile(np.nansum(node2attribute, axis=0), (n_nodes, 1))
    nb_size = np.dot(neighborhoods, nodes_not_nan.astype(int))[:, np.newaxis]
    in_nb = np.tile(nb_size, (1, n_attr))
    hits = np.dot(neighborhoods, np.where(~np.isnan(node2attribute), node2attribute, 0))
    with np.errstate(invalid='ignore', divide='ignore'):
        pvalues_pos = hypergeom.sf(hits - 1, pop, in_group, in_nb)
        if multiple_testing:                               # safe.py:599-605
            pvalues_pos = fdr_rows(pvalues_pos)
        nes = -np.log10(pvalues_pos)
    return {'pvalues_pos': pvalues_pos, 'nes': nes}


def pvalues_by_randomization(neighborhoods, node2attribute, neighborhood_score_type,
                             num_permutations, random_seed, attribute_sign, multiple_testing=False):
    """safe.py:496-554 without the sleep and the (broken) multiprocessing split."""
    ns = compute_neighborhood_score(neighborhoods, node2attribute, neighborhood_score_type)
    counts_neg, counts_pos = run_permutations(neighborhoods, node2attribute,
                                              neighborhood_score_type, num_permutations, random_seed)
    idx = np.isnan(ns)
    counts_neg[idx] = np.nan
    counts_pos[idx] = np.nan
    pvalues_neg = counts_neg / num_permutations
    pvalues_pos = counts_pos / num_permutations
    if multiple_testing:                                   # safe.py:536-542
        pvalues_neg = fdr_rows(pvalues_neg)
        pvalues_pos = fdr_rows(pvalues_pos)
    with np.errstate(invalid='ignore', divide='ignore'):
        nes_pos = -np.log10(np.where(pvalues_pos == 0, 1 / num_permutations, pvalues_pos))
        nes_neg = -np.log10(np.where(pvalues_neg == 0, 1 / num_permutations, pvalues_neg))
    if attribute_sign == 'highest':
        nes = nes_pos
    elif attribute_sign == 'lowest':
        nes = nes_neg
    else:
        nes = nes_pos - nes_neg
    return {'ns': ns, 'pvalues_neg': pvalues_neg, 'pvalues_pos': pvalues_pos, 'nes': nes}


def binarize(nes, enrichment_threshold):
    """safe.py:468-472."""
    idx = ~np.isnan(nes)
    nes_binary = np.zeros(nes.shape)
    nes_binary[idx] = np.abs(nes[idx]) > -np.log10(enrichment_threshold)
    return nes_binary, np.sum(nes_binary, axis=0)


def compute_pvalues(neighborhoods, node2attribute, enrichment_type='auto', neighborhood_score_type='sum',
                    background='attribute_file', num_permutations=1000, random_seed=None,
                    attribute_sign='both', enrichment_threshold=0.05, multiple_testing=False):
    """safe.py:432-472.  ``node2attribute`` is modified in place when background == 'network'
    exactly as the reference does (safe.py:449-451)."""
    if background == 'network':
        node2attribute[np.isnan(node2attribute)] = 0
    if wants_hypergeometric(node2attribute, enrichment_type):
        out = pvalues_by_hypergeom(neighborhoods, node2attribute, multiple_testing)
    else:
        out = pvalues_by_randomization(neighborhoods, node2attribute, neighborhood_score_type,
                                       num_permutations, random_seed, attribute_sign, multiple_testing)
    out['nes_binary'], out['num_neighborhoods_enriched'] = binarize(out['nes'], enrichment_threshold)
    return out


# ---------------------------------------------------------------------------
# The device stream of UNSEEDED runs (safepy_amd/csrc/rng.cpp, k_perms_device), restated for the tests.  It has no counterpart
# to reproduce in the reference -- random_seed=None seeds from OS entropy (safe.py:88, safe_extras.py:46) -- so what is pinned
# here is the product's own documented algorithm (a scatter shuffle: 64 random buckets, Fisher-Yates inside each, Philox4x32-10
# words named by counters, Lemire's unbiased bounded draw -- see device_stream_tables).  The statistical claims (uniform,
# independent rows) are tested separately.
# ---------------------------------------------------------------------------

def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 over uint64 arrays holding 32-bit values; returns the four output words."""
    m0, m1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = m0 * c0, m1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c0, c1, c2, c3


def _device_draw(q, b, i, key0, key1):
    rng_range = i + 1
    thresh = ((1 << 32) - rng_range) % rng_range
    w = _philox4x32_10([q], [(b << 16) | (i >> 2)], [0x5AFE], [0], key0, key1)
    m = int(w[i & 3][0]) * rng_range
    if (m & 0xFFFFFFFF) >= thresh:
        return m >> 32
    r = 0
    while True:
        w = _philox4x32_10([q], [(b << 16) | i], [0xFA11], [r], key0, key1)
        for t in range(4):
            m = int(w[t][0]) * rng_range
            if (m & 0xFFFFFFFF) >= thresh:
                return m >> 32
        r += 1


def device_stream_tables(n, movable, num_permutations, key):
    """int64 [P, n]: the composed tables safe_perms_create_device generates for `key` (a 64-bit integer).  Row q: every movable
    position e draws a bucket (six bits of byte e & 3 of word (e >> 2) & 3 of Philox counter (q, e >> 4, 0xB0C7, 0)); buckets
    keep their elements in ascending order and are shuffled in place by Fisher-Yates from the top with draws from counter
    (q, bucket << 16 | i >> 2, 0x5AFE, 0), word i & 3 (Lemire's multiply-shift; a rejected word falls back to the words of
    (q, bucket << 16 | i, 0xFA11, r)); the buckets laid end to end are the permutation."""
    movable = np.asarray(movable).astype(bool)
    mov = np.flatnonzero(movable)
    k = len(mov)
    key0, key1 = int(key) & 0xFFFFFFFF, (int(key) >> 32) & 0xFFFFFFFF
    out = np.tile(np.arange(n, dtype=np.int64), (num_permutations, 1))
    if k == 0:
        return out
    e = np.arange(k, dtype=np.uint64)
    n_blocks = (k + 15) // 16
    # the FIRST word of every Fisher-Yates step of every bucket, in bulk: steps i < bucket size <= k
    for q in range(num_permutations):
        blocks = np.stack(_philox4x32_10(np.full(n_blocks, q, dtype=np.uint64), np.arange(n_blocks, dtype=np.uint64),
                                         np.full(n_blocks, 0xB0C7, dtype=np.uint64), np.zeros(n_blocks, dtype=np.uint64), key0, key1), axis=1)
        word = blocks[(e >> np.uint64(4)).astype(np.int64), ((e >> np.uint64(2)) & np.uint64(3)).astype(np.int64)]
        bucket = ((word >> (np.uint64(8) * (e & np.uint64(3)))) & np.uint64(63)).astype(np.int64)
        a = []
        for b in range(64):
            xb = [int(v) for v in np.flatnonzero(bucket == b)]
            size = len(xb)
            if size >= 2:
                steps = np.arange(size, dtype=np.uint64)
                first = np.stack(_philox4x32_10(np.full(size, q, dtype=np.uint64), (np.uint64(b) << np.uint64(16)) | (steps >> np.uint64(2)),
                                                np.full(size, 0x5AFE, dtype=np.uint64), np.zeros(size, dtype=np.uint64), key0, key1), axis=1)
                first = first[np.arange(size), (steps & np.uint64(3)).astype(np.int64)]
                rng_range = steps + np.uint64(1)
                thresh = (np.uint64(1 << 32) - rng_range) % rng_range
                m = first * rng_range
                accepted = (m & np.uint64(0xFFFFFFFF)) >= thresh
                draws = (m >> np.uint64(32)).astype(np.int64)
                for i in range(size - 1, 0, -1):
                    j = int(draws[i]) if accepted[i] else _device_draw(q, b, i, key0, key1)
                    xb[i], xb[j] = xb[j], xb[i]
            a.extend(xb)
        out[q, mov] = mov[np.array(a, dtype=np.int64)]
    return out


# ---------------------------------------------------------------------------
# Legacy MT19937 permutation stream, restated (numpy legacy RandomState;
# SURVEY Appendix A.3).  Used to pin the product's host RNG against an
# independent statement AND against numpy itself.
# ---------------------------------------------------------------------------

class LegacyMT19937:
    def __init__(self, seed):
        mt = np.empty(624, dtype=np.uint64)
        s = int(seed) & 0xFFFFFFFF
        for i in range(624):
            mt[i] = s
            s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xFFFFFFFF
        self.mt = [int(v) for v in mt]
        self.pos = 624

    def _refill(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            v = mt[(i + 397) % 624] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[i] = v
        self.pos = 0

    def next_u32(self):
        if self.pos == 624:
            self._refill()
        y = self.mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def interval(self, top):
        """Uniform integer in [0, top] by masked rejection (legacy random_interval)."""
        if top == 0:
            return 0
        mask = top
        for sh in (1, 2, 4, 8, 16):
            mask |= mask >> sh
        while True:
            v = self.next_u32() & mask
            if v <= top:
                return v

    def permutation(self, values):
        a = list(values)
        for i in range(len(a) - 1, 0, -1):
            j = self.interval(i)
            a[i], a[j] = a[j], a[i]
        return a


# ---------------------------------------------------------------------------
# Consumers of nes_binary (safe.py:610-745), restated with the library calls the
# reference makes (networkx connected components, SciPy linkage / fcluster,
# pandas group-bys).  Pinned by tests/golden/domains.npz (real reference run).
# ---------------------------------------------------------------------------

def top_attributes(nes_binary, num_enriched, n_nodes, edge_u, edge_v, min_size=10):
    """safe.py:626-656, attribute_unimodality_metric='connectivity'.  Returns dict of arrays:
    top (bool), num_connected_components, num_large_connected_components, and a list
    size_connected_components (None or descending sizes)."""
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(n_nodes))
    g.add_edges_from(zip([int(u) for u in edge_u], [int(v) for v in edge_v]))
    m = nes_binary.shape[1]
    top = np.asarray(num_enriched) >= min_size
    num_cc = np.zeros(m, dtype=np.int64)
    num_large = np.zeros(m, dtype=np.int64)
    sizes = [None] * m
    for a in np.flatnonzero(top):
        nodes = [v for v in range(n_nodes) if nes_binary[v, a] > 0]
        comps = sorted(nx.connected_components(nx.subgraph(g, nodes)), key=len, reverse=True)
        sz = np.array([len(c) for c in comps])
        num_cc[a] = len(comps)
        sizes[a] = sz
        num_large[a] = np.sum(sz >= min_size)
    top = top & ~(num_cc > 1)
    return {'top': top, 'num_connected_components': num_cc, 'size_connected_components': sizes,
            'num_large_connected_components': num_large}


def domains(nes, nes_binary, top, distance_metric='jaccard', distance_threshold=0.75):
    """safe.py:672-705.  Returns (domain per attribute, node2domain sums [N, D] with their domain ids,
    primary_domain, primary_nes)."""
    import pandas as pd
    from scipy.cluster.hierarchy import linkage, fcluster
    top = np.asarray(top, dtype=bool)
    z = linkage(nes_binary[:, top].T, method='average', metric=distance_metric)
    dom_top = fcluster(z, np.max(z[:, 2] * distance_threshold), criterion='distance')
    dom = np.zeros(nes_binary.shape[1], dtype=np.int64)
    dom[top] = dom_top
    cols = pd.MultiIndex.from_arrays([np.arange(len(dom)), dom], names=[None, 'domain'])
    sums = pd.DataFrame(nes_binary, columns=cols).T.groupby(level='domain').sum().T
    t = sums.loc[:, 1:]
    t_max = t.max(axis=1)
    primary = t.idxmax(axis=1)
    primary[t_max == 0] = 0
    best = pd.DataFrame(nes, columns=cols).T.groupby(level='domain').max().T
    primary_nes = np.array([best.loc[r, c] for r, c in zip(primary.index.values, primary.values)])
    return dom, sums.columns.values.astype(np.int64), sums.values, primary.values.astype(np.int64), primary_nes


# ---------------------------------------------------------------------------
# Callers / data formats either side of the path, pinned by
# tests/golden/io.npz (real reference run: .scatter network, weighted edge lengths,
# read_attributes on text / gzip / DataFrame inputs).
# ---------------------------------------------------------------------------

def weighted_edge_lengths(xy, edge_u, edge_v, weight):
    """safe_io.py:311-333 with edge weights: the reference multiplies the pdist matrix by the
    adjacency matrix (entries = weights, zeros turned into NaN first, :325-328) and keeps the
    non-NaN entries (:330).  Returns per edge d(u,v) * w, NaN where the edge gets no length."""
    d = edge_lengths(xy, edge_u, edge_v)
    w = np.asarray(weight, dtype=np.float64).copy()
    w[w == 0] = np.nan
    return d * w


def pseudo_network_edges(xy, neighborhood_radius):
    """safe.py:302-309: dense pdist matrix, radius scaled by the extent of ALL coordinates
    (`node_coordinates.ravel()`), strict `<`; nx.from_numpy_array keeps every non-zero entry,
    self loops included.  Returns the undirected edge list as sorted (u <= v) pairs."""
    xy = np.asarray(xy, dtype=np.float64)
    d = squareform(pdist(xy, 'euclidean'))
    nr = neighborhood_radius * (np.max(xy.ravel()) - np.min(xy.ravel()))
    u, v = np.nonzero(np.triu(d < nr))
    return np.stack([u, v], axis=1).astype(np.int64)


def parse_attribute_text(path):
    """safe_io.py:358-368: the text / gzip branch up to the numeric label-indexed table.  The text
    to float conversion and the float32 down-cast rule are pandas' (third-party; same calls)."""
    import pandas as pd
    t = pd.read_csv(path, sep='\t', dtype={0: str})
    t = t.set_index(t.columns[0], drop=True)
    t = t.apply(pd.to_numeric, downcast='float', errors='coerce')
    names = [str(c) for c in t.columns]
    t.columns = np.arange(t.shape[1])
    return names, t


def align_attributes(table, node_label_order=None, fill_value=np.nan, mask_duplicates=False):
    """safe_io.py:380-410 restated with explicit loops instead of pandas' reindex: coerce to numeric,
    average rows sharing a label, then one output row per network node -- the file's row of that
    label, or `fill_value` when the label is not in the file; with mask_duplicates only one random
    node per label (first in a `np.random.permutation` order) keeps its values.
    Returns (node_label_order, matrix in the table's common float dtype)."""
    import pandas as pd
    table = table.apply(pd.to_numeric, errors='coerce')
    if not table.index.is_unique:
        table = table.groupby(table.index).mean()
    labels = list(table.index.values)
    values = table.to_numpy()
    if values.dtype not in (np.float32, np.float64):
        values = values.astype(np.float64)
    if node_label_order is None or len(node_label_order) == 0:
        node_label_order = labels
    where = {lab: i for i, lab in enumerate(labels)}
    out = np.full((len(node_label_order), values.shape[1]), fill_value, dtype=values.dtype)
    for i, lab in enumerate(node_label_order):
        if lab in where:
            out[i] = values[where[lab]]
    if mask_duplicates:
        idx = np.random.permutation(np.arange(len(node_label_order)))
        seen = set()
        for i in idx:
            lab = node_label_order[i]
            if lab in seen:
                out[i] = np.nan
            seen.add(lab)
    return list(node_label_order), out


def value_census(mat):
    """safe_io.py:426-429."""
    ok = mat[~np.isnan(mat)]
    return int(np.sum(np.isnan(mat))), int(np.sum(ok == 0)), int(np.sum(ok > 0)), int(np.sum(ok < 0))
