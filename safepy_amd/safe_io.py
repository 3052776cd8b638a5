"""Callers and data formats either side of the hot path.

Host-side mirror of the pieces of `safepy/safe_io.py` that feed `define_neighborhoods()` /
`compute_pvalues()`, with their data-parallel steps on the device:

  calculate_edge_lengths      safe_io.py:311-333   per-edge kernel instead of the N x N pdist matrix,
                                                   the dense adjacency and the Python ndenumerate loop
  load_network_from_scatter   safe_io.py:271-285   (host parse, pandas like the reference)
  euclidean_pseudo_network    safe.py:302-309      the `.scatter` pseudo-network: all-pairs distance +
                                                   threshold kernel, edges read back from the device CSR
  read_attributes             safe_io.py:336-430   parse with pandas like the reference; the alignment to
                                                   network node order (`reindex` + `.values`) and the value
                                                   census of the log run on the device, and the aligned
                                                   matrix can stay resident for compute_pvalues

Layouts (spring / Kamada-Kawai), Cytoscape / MATLAB loaders and plotting stay out of scope.
There is no CPU fallback: every function that computes needs a HIP device.
"""
import logging
import os
import pickle

import numpy as np

from . import backend as be


def _xy_in_node_order(G):
    x = np.array([v for _, v in G.nodes.data('x')], dtype=np.float64)
    y = np.array([v for _, v in G.nodes.data('y')], dtype=np.float64)
    return np.stack([x, y], axis=1) if len(x) else np.zeros((0, 2))


def get_node_coordinates(graph):
    """[N,2] (x, y) in node order: the `labels=[]` branch of safe_io.py:649-662."""
    from .safe import LayoutGraph
    if isinstance(graph, LayoutGraph):
        return graph.xy
    return _xy_in_node_order(graph)


def calculate_edge_lengths(G, verbose=True, device=0):
    """safepy/safe_io.py:311-333: sets edge attribute 'length' = Euclidean distance between the
    end points (times the edge's 'weight' when it has one -- the reference multiplies the distance
    matrix by `nx.adjacency_matrix(G)`, whose entries are the weights -- and edges of weight 0 get
    no length).  Node ids must be 0..N-1 in node order: the reference addresses edges by matrix
    index (`np.ndenumerate`), which only names the right edge under that numbering.
    Accepts a networkx graph or a `LayoutGraph`; returns it."""
    from .safe import LayoutGraph
    if verbose:
        logging.info('Calculating edge lengths...')
    ctx = be.Context.default(device)
    if isinstance(G, LayoutGraph):
        d = ctx.edge_lengths(G.xy, G.edge_u, G.edge_v) if G.edge_u.size else np.zeros(0)
        if G.weight is not None:
            d = d * G.weight
        G.length = d
        return G
    nodes = list(G)
    if nodes != list(range(len(nodes))):
        raise ValueError('calculate_edge_lengths: node ids must be 0..N-1 in node order '
                         '(the reference indexes edges by adjacency-matrix position, safe_io.py:330)')
    xy = _xy_in_node_order(G)
    edges = list(G.edges(data='weight', default=1))
    if not edges:
        return G
    eu = np.fromiter((e[0] for e in edges), dtype=np.int64, count=len(edges))
    ev = np.fromiter((e[1] for e in edges), dtype=np.int64, count=len(edges))
    w = np.array([e[2] for e in edges], dtype=np.float64)
    d = ctx.edge_lengths(xy, eu, ev)
    with np.errstate(invalid='ignore'):
        val = d * w                                    # np.multiply(node_distances, adjacency_matrix), safe_io.py:328
    keep = (w != 0) & ~np.isnan(val)                   # zero entries of the adjacency become NaN and are dropped (:326, :330)
    for u, v, x, k in zip(eu.tolist(), ev.tolist(), val.tolist(), keep.tolist()):
        if k:
            G[u][v]['length'] = x
    return G


def load_network_from_gpickle(filename, verbose=True):
    """safepy/safe_io.py:124-130."""
    with open(os.path.expanduser(filename), 'rb') as f:
        return pickle.load(f)


def load_network_from_scatter(filename, node_key_attribute='key', verbose=True):
    """safepy/safe_io.py:271-285: tab-separated file with a header line and four columns
    (key, x, y, label) -> an edgeless networkx graph, nodes 0..N-1 with those attributes."""
    import networkx as nx
    import pandas as pd
    if verbose:
        logging.info('Loading the file of node coordinates...')
    scatter = pd.read_csv(os.path.expanduser(filename), sep='\t')
    scatter.columns = ['key', 'x', 'y', 'label']
    G = nx.Graph()
    G.add_nodes_from([(i, row) for i, row in scatter.T.to_dict().items()])
    return G


def euclidean_pseudo_network(graph, neighborhood_radius, device=0, as_networkx=True):
    """The pseudo-network `load_network` attaches to `.scatter` inputs (safepy/safe.py:302-309):
    nodes closer than `neighborhood_radius * (max(coords) - min(coords))` -- the extent is taken
    over x AND y values together (safe.py:306) -- are connected; every node also gets a self loop
    (distance 0 < radius), and every edge weight 1.0, as `nx.from_numpy_array` produces.
    The all-pairs distances and the threshold run in the K1 kernel; the edge list is read back
    from the device CSR (u <= v).  as_networkx=False returns a `LayoutGraph` (arrays only)."""
    from .safe import LayoutGraph
    xy = np.ascontiguousarray(get_node_coordinates(graph), dtype=np.float64)
    n = xy.shape[0]
    nr = neighborhood_radius * (np.max(xy.ravel()) - np.min(xy.ravel()))
    ctx = be.Context.default(device)
    nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
    try:
        row_ptr, col = nbr.csr()
    finally:
        nbr.close()
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_ptr))
    col = col.astype(np.int64)
    upper = col >= rows
    eu, ev = rows[upper], col[upper]
    if not as_networkx:
        return LayoutGraph(xy, eu, ev, weight=np.ones(eu.size))
    import networkx as nx
    G = nx.Graph()
    G.add_nodes_from(range(n))
    G.add_edges_from(zip(eu.tolist(), ev.tolist()), weight=1.0)
    return G


# ------------------------------------------------------------------------------------------------
# read_attributes
# ------------------------------------------------------------------------------------------------
def _numeric_table_from_text(path):
    """A tab-separated attribute file (first column = node labels, header = attribute names) as a
    label-indexed frame of floats.  Non-numeric cells become NaN and every column is stored in the
    narrowest float type that holds it -- which is why matrices loaded from text are float32
    (safe_io.py:358-365)."""
    import pandas as pd
    raw = pd.read_csv(path, sep='\t', dtype={0: str})
    raw = raw.set_index(raw.columns[0], drop=True)
    return raw.apply(pd.to_numeric, downcast='float', errors='coerce')


def _parse_attribute_source(attribute_file):
    """(attributes frame [id, name], label-indexed numeric table) from a `.txt` / `.gz` path or a
    DataFrame indexed by node label -- the input half of read_attributes (safe_io.py:338-388).  The
    text-to-float conversion and the averaging of repeated labels are pandas' own, as in the
    reference; columns are renamed 0..M-1 for files and kept for DataFrames."""
    import pandas as pd
    if isinstance(attribute_file, pd.DataFrame):
        table, names = attribute_file, attribute_file.columns
    elif isinstance(attribute_file, str):
        path = os.path.expanduser(attribute_file)
        ext = os.path.splitext(path)[1]
        if ext == '.mat':
            raise NotImplementedError('MATLAB attribute files (safe_io.py:346-356) are out of scope; '
                                      'pass a .txt / .gz file or a DataFrame')
        if ext not in ('.txt', '.gz'):
            raise ValueError("Only attribute files with the following extensions are accepted: .mat, .txt, .gz.")
        table = _numeric_table_from_text(path)
        names = table.columns
        table.columns = np.arange(table.shape[1])
    else:
        raise ValueError('attribute_file must be a path or a pandas DataFrame, got %s' % type(attribute_file))
    attributes = pd.DataFrame({'id': np.arange(len(names)), 'name': pd.Index(names).astype(str)})
    table = table.apply(pd.to_numeric, errors='coerce')
    if table.index.has_duplicates:
        logging.info('\nThe attribute file contains multiple values for the same labels. Their values will be averaged.')
        table = table.groupby(level=0).mean()
    return attributes, table


def read_attributes_device(attribute_file='', node_label_order=None, mask_duplicates=False, fill_value=np.nan,
                           verbose=True, device=0):
    """`read_attributes` that also returns the device-resident aligned matrix:
    (attributes, node_label_order, node2attribute, backend.Attributes).  The caller owns the handle."""
    import pandas as pd
    attributes, table = _parse_attribute_source(attribute_file)
    if node_label_order is None or len(node_label_order) == 0:       # `if not node_label_order`, safe_io.py:390
        node_label_order = table.index.values
    node_label_in_file = table.index.values
    known = set(node_label_order)
    node_label_not_mapped = [x for x in node_label_in_file if x not in known]     # :394 (a set instead of a list scan)

    # ---- the alignment, safe_io.py:396 `reindex(index=node_label_order, fill_value=...)` + :412 `.values`
    values = table.to_numpy()
    if values.dtype not in (np.float32, np.float64):
        values = values.astype(np.float64)            # integer / mixed tables: the reference's reindex with NaN upcasts too
    row_map = table.index.get_indexer(pd.Index(node_label_order)).astype(np.int64)      # -1 = label not in the file
    if mask_duplicates:                               # :399-409: one random node per label keeps its values
        idx = np.random.permutation(np.arange(len(row_map)))
        mask_dups = pd.Index(node_label_order)[idx].duplicated(keep='first')
        logging.info('\nThe network contains %d nodes with duplicate labels. '
                     'Only one random node per label will be considered. '
                     'The attribute values of all other nodes will be set to NaN.' % mask_dups.sum())
        row_map[idx[mask_dups]] = -2
    ctx = be.Context.default(device)
    if values.shape[0] == 0:                          # empty file: every node is "not in the file"
        values = np.full((1, values.shape[1]), fill_value, dtype=values.dtype)
    # memory order of the result: whatever pandas' `.values` yields for this frame (Fortran for a frame
    # that is one block per dtype straight from the parser, C after a group-by) -- a two-row probe of
    # the same call tells, since the order depends on the block structure, not on the row count
    probe = table.iloc[:2].reindex(index=list(node_label_order[:2]), fill_value=fill_value).values
    order = 'C' if (probe.flags['C_CONTIGUOUS'] and not probe.flags['F_CONTIGUOUS']) else 'F'
    attr, node2attribute = be.Attributes.reindexed(ctx, values, row_map, fill_value=fill_value, order=order)

    if verbose:                                       # the summary of safe_io.py:414-431; the value census is one device pass
        n_file, n_attr, n_lost = len(node_label_in_file), attributes.shape[0], len(node_label_not_mapped)
        logging.info('\nAttribute data provided: %d labels x %d attributes' % (n_file, n_attr))
        if n_lost:
            shown = [str(x) for x in node_label_not_mapped[:3]]
            logging.info('%s and %d other labels in the attribute file were not found in the network.'
                         % (', '.join(shown), n_lost - len(shown)))
        logging.info('\nAttribute data mapped onto the network: %d labels x %d attributes' % (n_file - n_lost, n_attr))
        for what, count in zip(('NaNs', 'zeros', 'positives', 'negatives'), attr.value_counts()):
            logging.info('Values: %d %s' % (count, what))
    return attributes, node_label_order, node2attribute, attr


def read_attributes(attribute_file='', node_label_order=None, mask_duplicates=False, fill_value=np.nan, verbose=True,
                    device=0):
    """safepy/safe_io.py:336-430.  Returns (attributes, node_label_order, node2attribute) like the
    reference; node2attribute is [len(node_label_order), M] in the table's float dtype (f32 for
    text files whose columns pandas can down-cast, else f64), Fortran order like pandas' `.values`
    of a single-dtype frame."""
    attributes, node_label_order, node2attribute, attr = read_attributes_device(
        attribute_file=attribute_file, node_label_order=node_label_order, mask_duplicates=mask_duplicates,
        fill_value=fill_value, verbose=verbose, device=device)
    attr.close()
    return attributes, node_label_order, node2attribute
