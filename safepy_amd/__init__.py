"""safepy_amd -- MI355X (gfx950) implementation of the SAFE hot path.

`SAFE.define_neighborhoods()` / `SAFE.compute_pvalues()` (+ `compute_node_distances()`)
with the reference's signatures and array layouts, computed by hand-written HIP kernels
in libsafe_hip.so through the C ABI of include/safe_hip.h.  Importing the package loads
the shared library and fails loudly if it is missing: there is no CPU fallback.
"""
from . import _lib                       # noqa: F401  (loads libsafe_hip.so or raises)
from ._lib import SafeHipError, LIB_PATH, device_count
from .backend import Context, Neighborhoods, Attributes, Permutations
from .safe import SAFE, LayoutGraph
from .safe_extras import compute_neighborhood_score, run_permutations
from .safe_io import calculate_edge_lengths, read_attributes, load_network_from_scatter, euclidean_pseudo_network

__all__ = ['SAFE', 'LayoutGraph', 'compute_neighborhood_score', 'run_permutations', 'calculate_edge_lengths',
           'read_attributes', 'load_network_from_scatter', 'euclidean_pseudo_network', 'Context', 'Neighborhoods',
           'Attributes', 'Permutations', 'SafeHipError', 'LIB_PATH', 'device_count']
