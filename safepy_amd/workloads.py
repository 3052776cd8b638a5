"""Seeded synthetic surrogates of the BASELINE.json configurations.

safe-data (Costanzo 2016 network, GO-BP matrix) is not available offline, so the bench and
the full-size tests run on surrogates with the same shape and statistics.  Pure NumPy;
used by bench.py and tests, never by the compute path.
"""
import numpy as np


def clustered_layout(rng, n, n_blobs=8, spread=0.03, background=0.8):
    """Gaussian blobs + uniform background on the unit square (ragged neighborhood sizes like
    the Costanzo layout: a few dense clusters and a sparse periphery)."""
    centers = rng.uniform(0.1, 0.9, size=(n_blobs, 2))
    widths = spread * rng.uniform(0.5, 2.0, size=n_blobs)
    xy = rng.uniform(0.0, 1.0, size=(n, 2))
    in_blob = rng.uniform(size=n) >= background
    which = rng.integers(0, n_blobs, size=n)
    noise = rng.normal(size=(n, 2)) * widths[which][:, None]
    xy[in_blob] = (centers[which] + noise)[in_blob]
    return xy


def radius_edges(xy, target_edges, rng, reach=0.15):
    """Undirected edges: a uniform sample of `target_edges` node pairs closer than
    reach * x-range (long edges relative to the neighborhood radius, as in a spring layout,
    so weighted shortest-path balls are much smaller than Euclidean ones)."""
    from scipy.spatial import cKDTree
    pairs = cKDTree(xy).query_pairs(reach * np.ptp(xy[:, 0]), output_type='ndarray')
    keep = rng.choice(len(pairs), size=min(target_edges, len(pairs)), replace=False)
    keep.sort()
    return pairs[keep, 0].astype(np.int64), pairs[keep, 1].astype(np.int64)


def go_like_binary(rng, n, m, n_nan_rows, density=0.00986, dtype=np.float32, order='F'):
    """Binary annotation matrix with GO-BP-like statistics (tests/test_enrichments.py:32-45 of
    the reference): heavy-tailed term sizes, `n_nan_rows` all-NaN rows, overall 1-density
    ~0.986 % of the annotated rows; f32 Fortran order like the .txt.gz loader
    (safe_io.py:361,410)."""
    rows = n - n_nan_rows
    sizes = np.exp(rng.uniform(np.log(2.0), np.log(500.0), size=m))
    sizes = np.maximum(1, np.round(sizes * density * rows * m / sizes.sum())).astype(np.int64)
    sizes = np.minimum(sizes, rows)
    b = np.zeros((n, m), dtype=dtype, order=order)
    nan_rows = rng.choice(n, size=n_nan_rows, replace=False)
    live = np.setdiff1d(np.arange(n), nan_rows)
    for j in range(m):
        b[rng.choice(live, size=sizes[j], replace=False), j] = 1
    b[nan_rows, :] = np.nan
    return b


def costanzo_surrogate(seed=0, n=3971, m=4373, target_edges=28202, n_nan_rows=182):
    """Config 2/3 inputs: layout, edges (+ Euclidean 'length'), GO-like binary attributes.
    Tuned against the reference's known answers (tests/test_neighborhoods.py:25-41): default
    metric r=0.1 gives 38.0 +/- 54.8 neighbors per node (Costanzo: 37.5 +/- 56.7), euclidean
    r=0.1 gives 137 +/- 64 (Costanzo: 148 +/- 41)."""
    rng = np.random.default_rng(seed)
    xy = clustered_layout(rng, n)
    eu, ev = radius_edges(xy, target_edges, rng)
    dx = xy[eu, 0] - xy[ev, 0]
    dy = xy[eu, 1] - xy[ev, 1]
    length = np.sqrt(dx * dx + dy * dy)
    b = go_like_binary(rng, n, m, n_nan_rows)
    return {'xy': xy, 'edge_u': eu, 'edge_v': ev, 'length': length, 'attributes': b}


def uniform_layout(seed, n):
    return np.random.default_rng(seed).uniform(size=(n, 2))


def quantitative_attributes(seed, n, m, nan_row_frac=0.05, nan_frac=0.01, dtype=np.float64):
    """Config 5 style: N(0,1) values, 5 % all-NaN rows, 1 % scattered NaNs."""
    rng = np.random.default_rng(seed)
    b = rng.normal(size=(n, m)).astype(dtype)
    b[rng.uniform(size=(n, m)) < nan_frac] = np.nan
    b[rng.choice(n, size=int(nan_row_frac * n), replace=False)] = np.nan
    return b


def example3_scatter(path, seed=3, n=1586, nan_frac=0.12):
    """The shape of the reference's only published timing (examples/Example_3_Scatterplot_annotation.ipynb:73,104,147-153:
    `networks/YeastPhenome_UMAP_1586.scatter`, euclidean r = 0.06, ONE quantitative attribute, 10 000 permutations, 16 s):
    writes a 1586-node `.scatter` file (tab-separated key / x / y / label, safe_io.py:271-285) of a UMAP-like clustered
    layout to `path` and returns (keys, xy, attribute DataFrame indexed by key).  The attribute is a normalised-phenotype-like
    column: N(0, 1) values kept to 10 fractional bits -- every neighborhood sum is then exact in f64 whatever the order of
    summation, so the permutation counts of any two correct implementations are EQUAL, not just close -- with ~12 % NaN."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    xy = clustered_layout(rng, n, n_blobs=14, spread=0.025, background=0.35) * 20.0 - 10.0          # UMAP-like coordinates
    keys = np.array(['Y%s%03d%s' % ('ABCDEFGHIJKLMNOP'[i % 16], i // 16, 'WC'[i % 2]) for i in range(n)])
    labels = np.array(['G%04d' % i for i in range(n)])
    pd.DataFrame({'key': keys, 'x': xy[:, 0], 'y': xy[:, 1], 'label': labels}).to_csv(path, sep='\t', index=False)
    # (the loader parses the decimal text back: use what it will read)
    xy = pd.read_csv(path, sep='\t')[['x', 'y']].to_numpy(dtype=np.float64)
    values = np.round(rng.normal(size=n) * 1024.0) / 1024.0
    values[rng.uniform(size=n) < nan_frac] = np.nan
    att = pd.DataFrame({'NPV hap alpha | growth (spot assay) | standard | YPG [3%] (surrogate)': values}, index=pd.Index(keys, name='Gene systematic name'))
    return keys, xy, att
