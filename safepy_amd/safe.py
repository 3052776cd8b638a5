"""`SAFE` -- host-side mirror of the reference class for the hot path.

Same constructor signature, attribute names, method names, kwargs, array layouts and
error behaviour as `safepy.safe.SAFE` (safepy/safe.py:37-608) for
`define_neighborhoods()` / `compute_pvalues()` (+ the additive `compute_node_distances()`);
the arithmetic runs in libsafe_hip.so on an MI355X.  File loaders, layouts, plotting,
domains and output writers of the reference are out of scope.
"""
import configparser
import logging
import os
import sys

import numpy as np
import pandas as pd        # noqa: F401  at import, like the reference (safepy/safe.py:19): the first compute_pvalues() of a process
                           # used to pay for it -- 166 of its 178 ms

from . import backend as be

_DEFAULTS = {
    # the [DEFAULT] section of safepy/safe_default.ini:1-24, restated (the shipped .ini is not copied):
    # every key read_config looks up, with the reference's default value
    'safe_data': '',
    'networkfile': 'networks/Costanzo_Science_2016.gpickle',
    'annotationfile': 'attributes/hoepfner_movva_2014_doxorubucin.txt',
    'annotationsign': 'both',
    'randomSeed': '',
    'background': 'attribute_file',
    'nodeDistanceType': 'shortpath_weighted_layout',
    'neighborhoodRadius': '0.1',
    'neighborhoodRadiusType': 'diameter',
    'unimodalityType': 'connectivity',
    'groupDistanceType': 'jaccard',
    'groupDistanceThreshold': '0.75',
}


class LayoutGraph:
    """Minimal stand-in for the networkx graph the reference keeps in `self.graph`: node
    coordinates in node order plus an undirected edge list with optional per-edge
    'length' / 'weight'.  `SAFE` accepts either this or a networkx.Graph."""

    def __init__(self, xy, edge_u=None, edge_v=None, length=None, weight=None, keys=None, labels=None):
        self.xy = np.ascontiguousarray(xy, dtype=np.float64)
        if self.xy.ndim != 2 or self.xy.shape[1] != 2:
            raise ValueError('xy must be [N,2]')
        n = self.xy.shape[0]
        self.edge_u = np.zeros(0, np.int64) if edge_u is None else np.asarray(edge_u, dtype=np.int64)
        self.edge_v = np.zeros(0, np.int64) if edge_v is None else np.asarray(edge_v, dtype=np.int64)
        self.length = None if length is None else np.asarray(length, dtype=np.float64)
        self.weight = None if weight is None else np.asarray(weight, dtype=np.float64)
        self.keys = list(range(n)) if keys is None else list(keys)
        self.labels = [str(k) for k in self.keys] if labels is None else list(labels)

    def number_of_nodes(self):
        return self.xy.shape[0]


def _graph_arrays(graph):
    """(xy [N,2] in node order, edge_u, edge_v, length or None, weight or None) from a
    LayoutGraph or a networkx graph.  Follows the reference's access pattern: x/y via
    graph.nodes.data (safe.py:390-396), edges indexed by node id (safe.py:412-415)."""
    if isinstance(graph, LayoutGraph):
        return graph.xy, graph.edge_u, graph.edge_v, graph.length, graph.weight
    if hasattr(graph, 'is_directed') and graph.is_directed():
        raise NotImplementedError('directed graphs are not supported (the reference networks are undirected)')
    n = graph.number_of_nodes()
    x = np.array([v for _, v in graph.nodes.data('x')], dtype=np.float64)
    y = np.array([v for _, v in graph.nodes.data('y')], dtype=np.float64)
    xy = np.stack([x, y], axis=1) if n else np.zeros((0, 2))
    eu, ev, el, ew = [], [], [], []
    has_len = has_w = False
    for u, v, data in graph.edges(data=True):
        eu.append(u)
        ev.append(v)
        # networkx _weight_function: data.get(weight, 1) (weighted.py:78)
        el.append(data.get('length', 1))
        ew.append(data.get('weight', 1))
        has_len = has_len or ('length' in data)
        has_w = has_w or ('weight' in data)
    eu = np.asarray(eu, dtype=np.int64) if eu else np.zeros(0, np.int64)
    ev = np.asarray(ev, dtype=np.int64) if ev else np.zeros(0, np.int64)
    length = np.asarray(el, dtype=np.float64) if has_len else None
    weight = np.asarray(ew, dtype=np.float64) if has_w else None
    return xy, eu, ev, length, weight


class _DeviceResult:
    """An [n, m] float64 result of compute_pvalues still on the device: copied to the host the first
    time the attribute is read (see `_LazyArray`).  Owns its device buffer."""

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, shape

    def get(self, into=None):
        host = self.buf.download(self.shape, out=into)
        self.buf.free()
        return host

    def drop(self):
        self.buf.free()


def _local_only_refcount():
    probe = np.empty(1)
    return sys.getrefcount(probe)


_LOCAL_ONLY_REFCOUNT = _local_only_refcount()


class _LazyArray:
    """Descriptor behind SAFE.ns / pvalues_neg / pvalues_pos / nes / nes_binary.  To the caller these
    are plain attributes holding host `float64 [N, M]` arrays (or None) exactly as in the reference
    (safe.py:530-554, 596-608, 468-472); compute_pvalues() leaves the matrices on the device and the
    copy (139 MB each at 3971 x 4373, 1.6 GB each at 20 000 x 10 000 -- ten to a hundred times the
    compute) happens on the first read of each one, so results that are never looked at are never
    moved.  `SAFE.lazy_outputs = False` restores the eager copies.

    Host arrays are recycled: when a result this descriptor handed out is replaced (the next compute_pvalues(), or the caller
    setting the attribute to None) the instance keeps the array, and the next read of the same shape copies into it -- IF nobody
    else holds a reference to it (sys.getrefcount), so an array the caller kept (`old = sf.nes`) is never written again.  A
    fresh 139 MB array costs 34 000 page faults on its first write (the copy runs at 35 GB/s) and as much again when it is
    freed; into resident pages the same copy runs at the link's 56 GB/s."""

    def __init__(self, name):
        self.slot = '_r_' + name
        self.made = '_made_' + name          # the host array this descriptor produced last (None: the caller's own value)
        self.spare = '_spare_' + name        # a produced array that was replaced: may be written again if no one else holds it

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        d = obj.__dict__
        v = d.get(self.slot)
        if isinstance(v, _DeviceResult):
            into = d.pop(self.spare, None)
            # (no reference outside this function: the count a fresh local array shows on THIS interpreter, measured once --
            # 2 on CPython 3.10 (local name + getrefcount's argument), possibly 1 where references are borrowed)
            if not (isinstance(into, np.ndarray) and into.shape == tuple(v.shape) and into.dtype == np.float64
                    and into.flags.c_contiguous and into.flags.owndata and sys.getrefcount(into) == _LOCAL_ONLY_REFCOUNT):
                into = None
            v = v.get(into)
            del into
            d[self.slot] = v
            d[self.made] = v
        return v

    def __set__(self, obj, value):
        d = obj.__dict__
        old = d.get(self.slot)
        if isinstance(old, _DeviceResult):
            old.drop()
        elif old is not None and old is d.get(self.made):
            d[self.spare] = old                                   # ours: kept for the next read of this attribute
        d[self.made] = None
        d[self.slot] = value


class SAFE:
    """Defines an instance of SAFE analysis (hot path only); see module docstring."""

    ns = _LazyArray('ns')
    pvalues_neg = _LazyArray('pvalues_neg')
    pvalues_pos = _LazyArray('pvalues_pos')
    nes = _LazyArray('nes')
    nes_binary = _LazyArray('nes_binary')

    def __init__(self, path_to_ini_file='', path_to_safe_data=None, verbose=True, device=0):
        self.verbose = verbose
        self.default_config = None
        self.path_to_safe_data = path_to_safe_data
        self.path_to_network_file = None
        self.view_name = None
        self.path_to_attribute_file = None

        self.graph = None
        self.graph_euclidean = None
        self.node_key_attribute = 'label_orf'

        self.attributes = None
        self.nodes = None
        self.node2attribute = None
        self.num_nodes_per_attribute = None
        self.attribute_sign = 'both'

        self.node_distance_metric = 'shortpath_weighted_layout'
        self.neighborhood_radius_type = None
        self.neighborhood_radius = None

        self.background = 'attribute_file'
        self.num_permutations = 1000
        self.multiple_testing = False
        self.neighborhood_score_type = 'sum'
        self.enrichment_type = 'auto'
        self.enrichment_threshold = 0.05
        self.enrichment_max_log10 = 16
        self.attribute_enrichment_min_size = 10
        self.random_seed = None

        self.ns = None
        self.pvalues_neg = None
        self.pvalues_pos = None
        self.nes = None
        self.nes_threshold = None
        self.nes_binary = None

        self.graph_euclidean = None      # Euclidean pseudo-network of .scatter inputs (safe.py:67, 302-309), if any
        self.attribute_unimodality_metric = 'connectivity'
        self.attribute_distance_metric = 'jaccard'
        self.attribute_distance_threshold = 0.75
        self.domains = None
        self.node2domain = None
        self.output_dir = ''

        # device state (never pickled)
        self.device = device
        self.lazy_outputs = True         # result matrices stay on the device until first read (_LazyArray)
        self._nbr = None                 # backend.Neighborhoods matching _neighborhoods_host / lazily downloaded
        self._neighborhoods_host = None
        self._node_distances = None
        self._pending_binary = None
        self._attr_dev = None            # resident node2attribute (load_attributes(keep_on_device=True))
        self._attr_dev_host = None

        self.read_config(path_to_ini_file, path_to_safe_data=self.path_to_safe_data)
        self.validate_config()

    # ------------------------------------------------------------------ config ----
    def read_config(self, path_to_ini_file, path_to_safe_data=None):
        """safepy/safe.py:116-188.  Three sources, later wins: the built-in defaults, the user's INI
        (sections 'Input files' and 'Analysis parameters'), the constructor's `path_to_safe_data`."""
        parser_options = dict(allow_no_value=True, comment_prefixes=('#', ';', '{'), inline_comment_prefixes='#')
        builtin = configparser.ConfigParser(**parser_options)
        builtin.read_dict({'DEFAULT': _DEFAULTS})
        self.default_config = builtin['DEFAULT']
        user = configparser.ConfigParser(defaults=self.default_config, **parser_options)
        if path_to_ini_file:
            user.read(path_to_ini_file)

        def option(section, key):
            # a section the user's file lacks answers with the defaults
            if not user.has_section(section):
                user.add_section(section)
            return user.get(section, key)

        # -- input files: relative to the data folder when there is one, else taken as given (safe.py:149-165)
        data_dir = path_to_safe_data if path_to_safe_data is not None else (option('Input files', 'safe_data') or None)
        self.path_to_safe_data = data_dir
        network_file, attribute_file = option('Input files', 'networkfile'), option('Input files', 'annotationfile')
        if data_dir is not None:
            assert data_dir.endswith('/'), "path_to_safe_data should end with '/' (it is joined with the file names)"
            network_file, attribute_file = os.path.join(data_dir, network_file), os.path.join(data_dir, attribute_file)
        self.path_to_network_file, self.path_to_attribute_file = network_file, attribute_file
        self.attribute_sign = option('Input files', 'annotationsign')

        # -- analysis parameters (safe.py:169-184)
        self.background = option('Analysis parameters', 'background')
        self.node_distance_metric = option('Analysis parameters', 'nodeDistanceType')
        self.neighborhood_radius_type = option('Analysis parameters', 'neighborhoodRadiusType')
        self.neighborhood_radius = float(option('Analysis parameters', 'neighborhoodRadius'))
        try:
            self.random_seed = int(option('Analysis parameters', 'randomSeed'))
        except (ValueError, TypeError):                    # empty = unseeded
            self.random_seed = None
        self.attribute_unimodality_metric = option('Analysis parameters', 'unimodalityType')
        self.attribute_distance_metric = option('Analysis parameters', 'groupDistanceType')
        self.attribute_distance_threshold = float(option('Analysis parameters', 'groupDistanceThreshold'))

        # the reference falls back on its package folder (safe.py:186-188); here: this package's
        self.output_dir = os.path.dirname(path_to_ini_file) if path_to_ini_file else ''
        if not self.output_dir:
            self.output_dir = os.path.dirname(os.path.abspath(__file__))

    def validate_config(self):
        """safepy/safe.py:190-235: invalid option -> restore the default, raise ValueError."""
        if self.background not in ['attribute_file', 'network']:
            bad, self.background = self.background, self.default_config.get('background')
            raise ValueError('%s is not a valid setting for background. '
                             'Valid options are: attribute_file, network.' % bad)
        if self.node_distance_metric not in ['euclidean', 'shortpath', 'shortpath_weighted_layout']:
            bad, self.node_distance_metric = self.node_distance_metric, self.default_config.get('nodeDistanceType')
            raise ValueError('%s is not a valid setting for node_distance_metric. '
                             'Valid options are: euclidean, shortpath, shortpath_weighted_layout' % bad)
        if self.attribute_sign not in ['highest', 'lowest', 'both']:
            bad, self.attribute_sign = self.attribute_sign, self.default_config.get('annotationsign')
            raise ValueError('%s is not a valid setting for attribute_sign. '
                             'Valid options are: highest, lowest, both' % bad)
        if not isinstance(self.num_permutations, int) or (self.num_permutations < 10):
            self.num_permutations = 1000
            raise ValueError('num_permutations must be an integer equal or greater than 10.')
        if not isinstance(self.enrichment_threshold, float) or (self.enrichment_threshold <= 0) \
                or (self.enrichment_threshold >= 1):
            self.enrichment_threshold = 0.05
            raise ValueError('enrichment_threshold must be in the (0,1) range.')
        if not isinstance(self.enrichment_max_log10, (int, float)):
            self.enrichment_max_log10 = 16
            raise ValueError('enrichment_max_log10 must be a number.')
        if not isinstance(self.attribute_enrichment_min_size, int) or (self.attribute_enrichment_min_size < 2):
            self.attribute_enrichment_min_size = 10
            raise ValueError('attribute_enrichment_min_size must be an integer equal or greater than 2.')
        if not isinstance(self.attribute_distance_threshold, float) or (self.attribute_distance_threshold <= 0) \
                or (self.attribute_distance_threshold >= 1):
            self.attribute_distance_threshold = 0.75
            raise ValueError('attribute_enrichment_min_size must be a float number in the (0,1) range.')

    # pickling drops device handles (the reference pickles the whole object, safe.py:237-242)
    def __getstate__(self):
        for name in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
            getattr(self, name)              # results still on the device come to the host first
        nd = self._node_distances
        if isinstance(nd, tuple) and nd[0] == 'device-shortpath':
            self._node_distances = ('dense-shortpath', nd[1].distances())
        state = self.__dict__.copy()
        if state.get('_neighborhoods_host') is None and state.get('_nbr') is not None:
            state['_neighborhoods_host'] = self._nbr.to_dense()
        for key in [k for k in state if k.startswith('_spare_') or k.startswith('_made_')]:
            del state[key]                   # recycled host arrays and their bookkeeping are not part of the object's value
        state['_nbr'] = None
        state['_attr_dev'] = None
        state['_attr_dev_host'] = None
        state['default_config'] = dict(self.default_config) if self.default_config is not None else None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        if isinstance(self.default_config, dict):
            cp = configparser.ConfigParser()
            cp.read_dict({'DEFAULT': self.default_config})
            self.default_config = cp['DEFAULT']

    # ------------------------------------------------------------------ inputs ----
    def load_network(self, **kwargs):
        """safepy/safe.py:244-324 for in-memory graphs (`graph=` / `network_file=` a networkx.Graph
        with node attributes x, y -- and edge attribute 'length' for the default metric -- or a
        `LayoutGraph`), `.gpickle` files and `.scatter` files (with their Euclidean pseudo-network,
        safe.py:296-309, built on the device).  The reference's other file loaders and layouts
        (safe_io.py:30-268, 288-308) are out of scope.  Sets self.graph, self.graph_euclidean (for
        .scatter) and self.nodes (safe.py:311-324)."""
        import pandas as pd
        from . import safe_io
        if 'network_file' in kwargs and isinstance(kwargs['network_file'], str):
            if self.path_to_safe_data is None:
                self.path_to_network_file = kwargs['network_file']
            else:
                self.path_to_network_file = os.path.join(self.path_to_safe_data, kwargs['network_file'])
        if 'view_name' in kwargs:
            self.view_name = kwargs['view_name']
        if 'node_key_attribute' in kwargs:
            self.node_key_attribute = kwargs['node_key_attribute']
        self.validate_config()
        graph = kwargs.get('graph', kwargs.get('network_file'))
        self.graph_euclidean = None
        if graph is None and self.path_to_network_file and os.path.exists(self.path_to_network_file):
            graph = self.path_to_network_file          # the configured network (INI networkfile / safe_data), safe.py:263-264
        if graph is None:
            raise NotImplementedError('safepy_amd.SAFE.load_network needs graph=<networkx.Graph | LayoutGraph> or '
                                      'network_file=<.gpickle | .scatter>; the default safe-data network is not bundled')
        if isinstance(graph, str):
            path = self.path_to_network_file
            assert os.path.exists(path), path
            suffixes = [x for x in os.path.basename(path).split('.')[1:]]
            ext = '.' + suffixes[0] if suffixes else ''
            if self.verbose:
                logging.info('Loading network from %s' % path)
            if ext == '.gpickle':
                graph = safe_io.load_network_from_gpickle(path, verbose=self.verbose)
            elif ext == '.scatter':
                graph = safe_io.load_network_from_scatter(path, node_key_attribute=self.node_key_attribute,
                                                          verbose=self.verbose)
                self.graph_euclidean = safe_io.euclidean_pseudo_network(
                    graph, self.neighborhood_radius, device=self.device,
                    as_networkx=kwargs.get('pseudo_network', 'networkx') == 'networkx')
            else:
                raise NotImplementedError('network files of type %r need the reference\'s loaders and layouts, which are '
                                          'out of scope for the hot-path build (supported: .gpickle, .scatter)' % ext)
        self.graph = graph
        self._invalidate_neighborhoods()
        if isinstance(graph, LayoutGraph):
            ids, keys, labels = list(range(graph.number_of_nodes())), list(graph.keys), list(graph.labels)
        else:
            key_list = dict(graph.nodes.data(self.node_key_attribute))
            key_list = {k: v for k, v in key_list.items() if v is not None}
            if not key_list:
                raise Exception('The specified node key attribute (%s) does not exist in this network. '
                                'Set node_key_attribute to one of the attributes the nodes carry.'
                                % self.node_key_attribute)
            for k, v in key_list.items():
                graph.nodes[k]['key'] = v
            label_list = {k: v for k, v in graph.nodes.data('label') if v is not None}
            ids, keys, labels = list(label_list.keys()), list(key_list.values()), list(label_list.values())
        self.nodes = pd.DataFrame(data={'id': ids, 'key': keys, 'label': labels})

    def load_attributes(self, **kwargs):
        """safepy/safe.py:334-367 over `read_attributes` (safe_io.py:336-430): `attribute_file=` a
        `.txt` / `.gz` path, a pandas DataFrame indexed by node key, or (additive) a ready [N,M]
        ndarray in node order; other kwargs (`mask_duplicates`, `fill_value`) are forwarded.
        The alignment to node order runs on the device.  `keep_on_device=True` (additive) keeps the
        aligned matrix resident for compute_pvalues(), which then skips the upload; the host
        `self.node2attribute` is made read-only in exchange (assign a new array to replace it)."""
        import pandas as pd
        from . import safe_io
        keep = bool(kwargs.pop('keep_on_device', False))
        self._drop_device_attributes()
        if 'attribute_file' in kwargs:
            src = kwargs.pop('attribute_file')
            if self.path_to_safe_data is None or isinstance(src, (pd.DataFrame, np.ndarray)):
                self.path_to_attribute_file = src
            elif isinstance(src, str):
                self.path_to_attribute_file = os.path.join(self.path_to_safe_data, src)
            else:
                raise ValueError(type(src))
        src = self.path_to_attribute_file
        if isinstance(src, str):
            assert os.path.exists(src), src
        self.validate_config()
        if isinstance(src, np.ndarray):
            self.node2attribute = src
            self.attributes = pd.DataFrame({'id': np.arange(src.shape[1]),
                                            'name': [str(j) for j in range(src.shape[1])]})
            if keep:
                self._attr_dev = be.Attributes.from_host(self._ctx(), src)
                self._attr_dev_host = src
                src.flags.writeable = False
            return
        if self.verbose and isinstance(src, str):
            logging.info('Loading attributes from %s' % src)
        self.attributes, _, self.node2attribute, attr = safe_io.read_attributes_device(
            node_label_order=self._node_keys(), verbose=self.verbose, attribute_file=src, device=self.device, **kwargs)
        if keep:
            self._attr_dev = attr
            self._attr_dev_host = self.node2attribute
            self.node2attribute.flags.writeable = False
        else:
            attr.close()

    def _drop_device_attributes(self):
        attr = self.__dict__.get('_attr_dev')
        if attr is not None:
            attr.close()
        self._attr_dev = None
        self._attr_dev_host = None

    def _resident_attributes(self):
        """The handle load_attributes(keep_on_device=True) left on the device, if it still mirrors
        self.node2attribute (same object, still read-only)."""
        attr = self.__dict__.get('_attr_dev')
        if attr is None:
            return None
        host = self.node2attribute
        if host is self._attr_dev_host and isinstance(host, np.ndarray) and not host.flags.writeable:
            return attr
        self._drop_device_attributes()
        return None

    def _node_keys(self):
        if isinstance(self.graph, LayoutGraph):
            return list(self.graph.keys)
        return [v for _, v in self.graph.nodes.data(self.node_key_attribute)]

    # ------------------------------------------------------- neighborhoods state ----
    @property
    def neighborhoods(self):
        """int64 [N,N] 0/1, C order (safe.py:387,430); downloaded from the device on first
        access after define_neighborhoods()."""
        if self._neighborhoods_host is None and self._nbr is not None:
            self._neighborhoods_host = self._nbr.to_dense()
        return self._neighborhoods_host

    @neighborhoods.setter
    def neighborhoods(self, value):
        self._invalidate_neighborhoods()
        self._neighborhoods_host = value

    def _invalidate_neighborhoods(self):
        nd = self._node_distances
        if isinstance(nd, tuple) and nd[0] == 'device-shortpath':     # still on the device, owned by the handle that goes away
            self._node_distances = ('dense-shortpath', nd[1].distances()) if nd[1] is self._nbr and nd[1].handle else None
        if self._nbr is not None:
            self._nbr.close()
        self._nbr = None
        self._neighborhoods_host = None

    def _ctx(self):
        return be.Context.default(self.device)

    def _device_neighborhoods(self):
        if self._nbr is None:
            if self._neighborhoods_host is None:
                raise RuntimeError('neighborhoods are not defined: call define_neighborhoods() first')
            self._nbr = be.Neighborhoods.from_dense(self._ctx(), self._neighborhoods_host)
            try:                          # a layout, when the graph has one, only orders the nodes on the device
                xy = _graph_arrays(self.graph)[0]
                if xy.shape == (self._nbr.n, 2) and np.isfinite(xy).all():
                    self._nbr.set_layout(xy)
            except Exception:
                pass
        return self._nbr

    @property
    def node_distances(self):
        """Shortest-path metrics: dict-of-dicts {source: {target: distance}} over reached
        pairs (safe.py:417).  Euclidean (additive, via compute_node_distances): f64 [N,N]."""
        nd = self._node_distances
        if isinstance(nd, tuple) and nd[0] == 'device-shortpath':     # the [N,N] f64 copy (126 MB at 3971 nodes) is made on first read
            nd = self._node_distances = ('dense-shortpath', nd[1].distances())
        if isinstance(nd, tuple) and nd[0] == 'dense-shortpath':
            dmat = nd[1]
            rows, cols = np.nonzero(np.isfinite(dmat))
            out = {int(s): {} for s in range(dmat.shape[0])}
            for s, t in zip(rows.tolist(), cols.tolist()):
                out[s][t] = float(dmat[s, t])
            self._node_distances = out
        return self._node_distances

    @node_distances.setter
    def node_distances(self, value):
        self._node_distances = value

    def _radius(self, xy):
        x = xy[:, 0]
        return self.neighborhood_radius * (np.max(x) - np.min(x))      # safe.py:390-391, 404-405

    def _override_neighborhood_settings(self, kwargs):
        if 'node_distance_metric' in kwargs:
            self.node_distance_metric = kwargs['node_distance_metric']
        if 'neighborhood_radius_type' in kwargs:
            self.neighborhood_radius_type = kwargs['neighborhood_radius_type']
        if 'neighborhood_radius' in kwargs:
            self.neighborhood_radius = kwargs['neighborhood_radius']
        self.validate_config()

    def _shortpath_inputs(self, xy, eu, ev, length, weight):
        n = xy.shape[0]
        if eu.size and (eu.min() < 0 or ev.min() < 0 or eu.max() >= n or ev.max() >= n):
            raise ValueError('shortest-path metrics index the neighborhood matrix by node id: ids must be 0..N-1')
        if self.node_distance_metric == 'shortpath_weighted_layout':
            w = length                                  # weight='length', missing -> 1 (networkx)
            cutoff = self._radius(xy)
        else:
            w = weight                                  # default weight attr 'weight', missing -> 1
            cutoff = self.neighborhood_radius          # safe.py:409
        return w, cutoff

    def define_neighborhoods(self, **kwargs):
        """safepy/safe.py:369-430.  kwargs: node_distance_metric, neighborhood_radius_type,
        neighborhood_radius (persisted on self).  Sets self.neighborhoods (and, for the
        shortest-path metrics, self.node_distances); returns None."""
        self._override_neighborhood_settings(kwargs)
        xy, eu, ev, length, weight = _graph_arrays(self.graph)
        ctx = self._ctx()
        if self.node_distance_metric != 'euclidean':
            self._node_distances = None          # replaced below (a copy still on the device is not fetched first)
        self._invalidate_neighborhoods()
        if self.node_distance_metric == 'euclidean':
            self._nbr = be.Neighborhoods.euclidean(ctx, xy, self._radius(xy))
        else:
            w, cutoff = self._shortpath_inputs(xy, eu, ev, length, weight)
            self._nbr = be.Neighborhoods.shortpath(ctx, xy.shape[0], eu, ev, w, cutoff, keep_distances=True)
            self._nbr.set_layout(xy)
            self._node_distances = ('device-shortpath', self._nbr)
        if self.verbose:
            num_neighbors = self._nbr.row_counts()
            logging.info('Node distance metric: %s' % self.node_distance_metric)
            logging.info('Neighborhood definition: %.2f x %s' % (self.neighborhood_radius, self.neighborhood_radius_type))
            logging.info('Number of nodes per neighborhood (mean +/- std): %.2f +/- %.2f'
                         % (np.mean(num_neighbors), np.std(num_neighbors)))

    def compute_node_distances(self, **kwargs):
        """Additive (named by the north star; absent from the reference at this commit):
        fills self.node_distances without touching self.neighborhoods.  Euclidean: dense
        f64 [N,N] == squareform(pdist(xy)) (safe.py:397).  Shortest-path metrics: the same
        dict-of-dicts define_neighborhoods stores (safe.py:417)."""
        self._override_neighborhood_settings(kwargs)
        xy, eu, ev, length, weight = _graph_arrays(self.graph)
        ctx = self._ctx()
        n = xy.shape[0]
        if self.node_distance_metric == 'euclidean':
            d_xy = ctx.alloc(xy.nbytes)
            d_out = ctx.alloc_f64(n, n)
            try:
                d_xy.upload(xy)
                ctx.euclidean_dense(d_xy.ptr, n, self._radius(xy), None, d_out.ptr)
                self._node_distances = d_out.download((n, n))
            finally:
                d_xy.free()
                d_out.free()
        else:
            w, cutoff = self._shortpath_inputs(xy, eu, ev, length, weight)
            nbr = be.Neighborhoods.shortpath(ctx, n, eu, ev, w, cutoff, keep_distances=True)
            self._node_distances = ('dense-shortpath', nbr.distances())
            nbr.close()

    # -------------------------------------------------------------- enrichment ----
    def compute_pvalues(self, **kwargs):
        """safepy/safe.py:432-472."""
        if 'how' in kwargs:
            self.enrichment_type = kwargs['how']
        if 'neighborhood_score_type' in kwargs:
            self.neighborhood_score_type = kwargs['neighborhood_score_type']
        if 'multiple_testing' in kwargs:
            self.multiple_testing = kwargs['multiple_testing']
        if 'background' in kwargs:
            self.background = kwargs['background']
        self.validate_config()

        resident = self._resident_attributes()
        if self.background == 'network':
            logging.info('Setting all null attribute values to 0. Using the network as background for enrichment.')
            if resident is not None:                                       # both copies, the host one stays read-only
                resident.nan_to_zero()
                self.node2attribute.flags.writeable = True
            if np.issubdtype(self.node2attribute.dtype, np.floating):      # (a uint8 / bool matrix has no missing values)
                self.node2attribute[np.isnan(self.node2attribute)] = 0     # in place, like safe.py:451
            if resident is not None:
                self.node2attribute.flags.writeable = False

        attr = resident if resident is not None else be.Attributes.from_host(self._ctx(), self.node2attribute)
        try:
            stats = attr.stats()
            if stats['max_nan_col'] / self.node2attribute.shape[0] > 0.5:
                logging.warning("WARNING: more than 50% of nodes in the network are set to NaN and "
                                "will be ignored for calculating enrichment.\n"
                                "Consider setting sf.background = 'network'.")
            self._pending_binary = None
            if (self.enrichment_type == 'hypergeometric') or \
                    ((self.enrichment_type == 'auto') and (stats['n_other'] == 0)):
                self.compute_pvalues_by_hypergeom(_attr=attr, **kwargs)
            else:
                self.compute_pvalues_by_randomization(_attr=attr, **kwargs)
        finally:
            if resident is None:
                attr.close()

        # safe.py:468-472 -- computed by the same kernels, from the same nes
        self.nes_binary, enriched = self._pending_binary
        self._pending_binary = None
        if self.attributes is None:
            self.attributes = pd.DataFrame({'id': np.arange(len(enriched)), 'name': [str(j) for j in range(len(enriched))]})
        self.attributes['num_neighborhoods_enriched'] = enriched

    def _result(self, buf, shape):
        """A finished [n, m] device buffer as the value of a result attribute: left on the device
        (lazy_outputs, the default) or copied to the host right away."""
        return _DeviceResult(buf, shape) if self.lazy_outputs else buf.download(shape)

    def compute_pvalues_by_randomization(self, _attr=None, **kwargs):
        """safepy/safe.py:474-554 (no 1 s sleep, no multiprocessing split: `processes` is
        accepted and ignored -- the reference's own split is broken at this commit)."""
        if kwargs:
            logging.warning('Current settings (possibly overwriting global ones):')
            for k in kwargs:
                logging.warning('\t%s=%s' % (k, str(kwargs[k])))
        logging.info('Using randomization to calculate enrichment...')
        if 'num_permutations' in kwargs:
            self.num_permutations = kwargs['num_permutations']
        self.validate_config()
        score_type = 'z-score' if self.neighborhood_score_type == 'z-score' else 'sum'

        ctx = self._ctx()
        nbr = self._device_neighborhoods()
        attr = _attr if _attr is not None else be.Attributes.from_host(ctx, self.node2attribute)
        n, m = attr.n, attr.m
        # random_seed=None (the default, like the reference's): the tables are generated on the device (backend.Permutations);
        # `device_stream_key` (None = OS entropy) makes such a run repeatable for tests and debugging
        perms = be.Permutations(ctx, n, attr.row_flags(), self.num_permutations, self.random_seed,
                                device_key=getattr(self, 'device_stream_key', None))
        bufs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
        try:
            be.randomization(ctx, nbr, attr, perms, score_type, self.attribute_sign, self.enrichment_threshold,
                             [b.ptr for b in bufs])
            if self.multiple_testing:                  # safe.py:536-542, then 546-554 and 468-472 on the adjusted values
                logging.info('Running FDR-adjustment of p-values...')
                be.fdr_adjust(ctx, n, m, self.num_permutations, self.attribute_sign, self.enrichment_threshold,
                              [b.ptr for b in bufs[1:]])
            enriched = bufs[5].download((m,))
            res = [self._result(b, (n, m)) for b in bufs[:5]]
            bufs = bufs[5:] if self.lazy_outputs else bufs       # handed over: the results own their buffers now
            self.ns, self.pvalues_neg, self.pvalues_pos, self.nes = res[:4]
            self._pending_binary = (res[4], enriched)
        finally:
            for b in bufs:
                b.free()
            perms.close()
            if _attr is None:
                attr.close()

    def compute_pvalues_by_hypergeom(self, _attr=None, **kwargs):
        """safepy/safe.py:556-608.  Sets pvalues_pos and nes only (ns / pvalues_neg untouched)."""
        if kwargs:
            if 'verbose' in kwargs:
                self.verbose = kwargs['verbose']
            if self.verbose:
                logging.warning('Overwriting global settings:')
                for k in kwargs:
                    logging.warning('\t%s=%s' % (k, str(kwargs[k])))
        self.validate_config()
        if self.verbose:
            logging.info('Using the hypergeometric test to calculate enrichment...')
        ctx = self._ctx()
        nbr = self._device_neighborhoods()
        attr = _attr if _attr is not None else be.Attributes.from_host(ctx, self.node2attribute)
        n, m = attr.n, attr.m
        bufs = [ctx.alloc_f64(n, m) for _ in range(3)] + [ctx.alloc_f64(m)]
        try:
            be.hypergeom(ctx, nbr, attr, self.enrichment_threshold, [b.ptr for b in bufs])
            if self.multiple_testing:                  # safe.py:599-605, then 608 and 468-472 on the adjusted values
                if self.verbose:
                    logging.info('Running FDR-adjustment of p-values...')
                be.fdr_adjust(ctx, n, m, 0, self.attribute_sign, self.enrichment_threshold,
                              [None] + [b.ptr for b in bufs])
            enriched = bufs[3].download((m,))
            res = [self._result(b, (n, m)) for b in bufs[:3]]
            bufs = bufs[3:] if self.lazy_outputs else bufs       # handed over: the results own their buffers now
            self.pvalues_pos, self.nes = res[:2]
            self._pending_binary = (res[2], enriched)
        finally:
            for b in bufs:
                b.free()
            if _attr is None:
                attr.close()

    # ------------------------------------------------------------------------------------
    # consumers of nes_binary
    # ------------------------------------------------------------------------------------
    def _graph_edges(self):
        """Edge list used for the connectivity of enriched nodes: self.graph, or the Euclidean
        pseudo-network of .scatter inputs when one is set (safe.py:643-645)."""
        g = getattr(self, 'graph_euclidean', None)
        if g is None:
            g = self.graph
        if isinstance(g, LayoutGraph):
            return g.edge_u, g.edge_v
        eu = np.fromiter((u for u, _ in g.edges()), dtype=np.int64, count=g.number_of_edges())
        ev = np.fromiter((v for _, v in g.edges()), dtype=np.int64, count=g.number_of_edges())
        return eu, ev

    def define_top_attributes(self, **kwargs):
        """safepy/safe.py:610-659.  kwargs: attribute_unimodality_metric, attribute_enrichment_min_size.
        Adds the columns 'top', 'num_connected_components', 'size_connected_components' (object: sizes in
        descending order) and 'num_large_connected_components' to self.attributes.  The connected
        components of all candidate attributes are found in one device call."""
        for option in ('attribute_unimodality_metric', 'attribute_enrichment_min_size'):
            if option in kwargs:
                setattr(self, option, kwargs[option])
        self.validate_config()
        min_size = self.attribute_enrichment_min_size
        if self.verbose:
            logging.info('Top attributes need >= %d enriched neighborhoods that form one region (%s)'
                         % (min_size, self.attribute_unimodality_metric))
        attrs = self.attributes
        attrs['top'] = (attrs['num_neighborhoods_enriched'] >= min_size).values           # requirement 1 (safe.py:628-629)

        if self.attribute_unimodality_metric == 'connectivity':                         # requirement 2 (safe.py:632-656)
            m_all = len(attrs)
            num_cc = np.zeros(m_all, dtype=np.int64)
            num_large = np.zeros(m_all, dtype=np.int64)
            sizes = np.empty(m_all, dtype=object)
            sizes[:] = None
            # like the reference, attribute index values are column positions of nes_binary
            cand = attrs.index.values[attrs['top'].values]
            if len(cand):
                eu, ev = self._graph_edges()
                n = self.nes_binary.shape[0]
                labels = be.enriched_components(self._ctx(), n, eu, ev, self.nes_binary[:, cand])
                pos_of = {a: i for i, a in enumerate(attrs.index.values)}
                for row, a in enumerate(cand):
                    lab = labels[row]
                    comp = np.bincount(lab[lab >= 0])
                    comp = np.sort(comp[comp > 0])[::-1]                  # (a few components, not the n bins, are sorted)
                    i = pos_of[a]
                    num_cc[i] = len(comp)
                    sizes[i] = comp
                    num_large[i] = int(np.sum(comp >= min_size))
            attrs['num_connected_components'] = num_cc
            attrs['size_connected_components'] = sizes
            attrs['num_large_connected_components'] = num_large
            attrs.loc[attrs['num_connected_components'] > 1, 'top'] = False              # safe.py:656
        if self.verbose:
            logging.info('Number of top attributes: %d' % np.sum(attrs['top']))

    def define_domains(self, **kwargs):
        """safepy/safe.py:661-713.  Average-linkage clustering of the top attributes on the distance
        between their binarised enrichment profiles (default: Jaccard, computed on the device in
        SciPy's condensed order; the linkage / fcluster calls are SciPy's, as in the reference), then
        every node's domain sums, primary domain and primary NES."""
        import pandas as pd
        from scipy.cluster.hierarchy import linkage, fcluster
        if 'attribute_distance_threshold' in kwargs:
            self.attribute_distance_threshold = kwargs['attribute_distance_threshold']
        self.validate_config()
        attrs = self.attributes
        top = attrs['top'].values.astype(bool)
        m = self.nes_binary[:, top].T
        if self.attribute_distance_metric == 'jaccard' and m.shape[0] >= 2:
            z = linkage(be.jaccard_condensed(self._ctx(), m), method='average')
        else:
            z = linkage(m, method='average', metric=self.attribute_distance_metric)
        max_d = np.max(z[:, 2] * self.attribute_distance_threshold)
        domains = fcluster(z, max_d, criterion='distance')
        attrs['domain'] = 0
        attrs.loc[attrs['top'], 'domain'] = domains

        # a node belongs to the domain holding most of the attributes it is enriched for (safe.py:693-698)
        dom = attrs['domain'].values
        ids = np.unique(dom)
        onehot = (dom[:, None] == ids[None, :]).astype(np.float64)
        sums = self.nes_binary @ onehot
        node2domain = pd.DataFrame(sums, columns=pd.Index(ids, name='domain'))
        real = ids >= 1
        t = sums[:, real]
        t_max = t.max(axis=1)
        primary = ids[real][np.argmax(t, axis=1)]                # first maximum, like DataFrame.idxmax
        primary = np.where(t_max == 0, 0, primary)
        node2domain['primary_domain'] = primary
        # the highest NES among the attributes of the primary domain (safe.py:703-705); NaNs are skipped.  Only the
        # (node, primary domain) pairs are evaluated: a domain's columns for the nodes that have it as primary domain
        # (all columns of every domain for every node -- domain 0 holds most of the matrix -- took 0.23 s at 3971 x 4373)
        if np.any(~np.isin(primary, ids)):
            raise KeyError(0)                                     # the reference's o.loc[row, 0] with no attribute outside the domains
        primary_nes = np.full(primary.shape[0], np.nan)
        with np.errstate(invalid='ignore'):
            for d in ids:
                rows = np.nonzero(primary == d)[0]
                if rows.size == 0:
                    continue
                block = self.nes[np.ix_(rows, np.nonzero(dom == d)[0])]
                nan = np.isnan(block)
                primary_nes[rows] = np.where(nan.all(axis=1), np.nan, np.where(nan, -np.inf, block).max(axis=1))
        node2domain['primary_nes'] = primary_nes
        self.node2domain = node2domain
        if self.verbose:
            per_domain = attrs.loc[attrs['domain'] > 0].groupby('domain')['id'].count()
            logging.info('Number of domains: %d (containing %d-%d attributes)'
                         % (len(np.unique(domains)), per_domain.min(), per_domain.max()))

    def trim_domains(self, **kwargs):
        """safepy/safe.py:715-745: a domain that is the primary domain of fewer than
        attribute_enrichment_min_size nodes is dissolved into domain 0; the survivors are renumbered
        0..D in ascending order and labelled with the five most frequent words of their attribute names
        (chop_and_filter, safe_io.py:735-745).  Sets self.domains."""
        import pandas as pd
        attrs, n2d = self.attributes, self.node2domain
        min_nodes = self.attribute_enrichment_min_size
        # how many nodes chose each domain id (ids are 0..D before trimming, safe.py:718-720)
        n_ids = attrs['domain'].nunique()
        votes = np.bincount(n2d['primary_domain'].to_numpy(dtype=np.int64), minlength=n_ids)[:n_ids]
        small = np.flatnonzero(votes < min_nodes)
        attrs.loc[attrs['domain'].isin(small), 'domain'] = 0
        n2d.loc[n2d['primary_domain'].isin(small), ['primary_domain', 'primary_nes']] = 0
        # dense renumbering of what is left (safe.py:729-734; the per-domain columns of node2domain keep their names there too)
        kept = np.sort(attrs['domain'].unique())
        stray = np.setdiff1d(n2d['primary_domain'].to_numpy(), kept)
        if stray.size:                                            # the reference's renumbering dictionary has no such key
            raise KeyError(int(stray[0]))
        attrs['domain'] = np.searchsorted(kept, attrs['domain'].to_numpy())
        n2d['primary_domain'] = np.searchsorted(kept, n2d['primary_domain'].to_numpy())
        labels = attrs.groupby('domain')['name'].apply(_domain_label)
        self.domains = pd.DataFrame({'id': np.arange(len(kept)), 'label': labels})
        if self.verbose:
            logging.info('Removed %d domains because they were the top choice for less than %d neighborhoods.'
                         % (len(small), min_nodes))


def _domain_label(names):
    """The five most frequent words of a domain's attribute names, most frequent first, ties in order
    of first appearance, without a few stop words (safe_io.py:735-745)."""
    import re
    from collections import Counter
    words = re.findall(r"[\w']+", names.str.cat(sep=' '))
    counts = Counter(words)
    ranked = sorted(counts, key=counts.get, reverse=True)
    skip = ('of', 'a', 'the', 'an', ',', 'via', 'to', 'into', 'from')
    return ', '.join([w for w in ranked if w not in skip][:5])
