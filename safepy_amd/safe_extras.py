"""MI355X counterparts of the reference's module-level hot-path functions
(safepy/safe_extras.py): same names, argument meaning and return values, computed by the
HIP kernels of libsafe_hip.so.  They can be monkey-patched into a safepy install
(INTEGRATION.md)."""
import numpy as np

from . import backend as be


def _nbr_handle(ctx, neighborhood2node):
    if isinstance(neighborhood2node, be.Neighborhoods):
        return neighborhood2node, False
    return be.Neighborhoods.from_dense(ctx, neighborhood2node), True


def compute_neighborhood_score(neighborhood2node, node2attribute, neighborhood_score_type, device=0):
    """safepy/safe_extras.py:6-33.  `neighborhood2node`: int [N,N] 0/1 membership (or a
    device-resident `backend.Neighborhoods`); `node2attribute`: f32/f64 [N,M], C or F order,
    NaN = missing; returns float64 [N,M].  Inputs are not modified."""
    if neighborhood_score_type not in ('sum', 'z-score'):
        # the reference silently falls through to the plain sum for any other string
        neighborhood_score_type = 'sum'
    ctx = be.Context.default(device)
    nbr, own = _nbr_handle(ctx, neighborhood2node)
    attr = be.Attributes.from_host(ctx, node2attribute)
    out = ctx.alloc_f64(attr.n, attr.m)
    try:
        be.score(ctx, nbr, attr, neighborhood_score_type, out.ptr)
        return out.download((attr.n, attr.m))
    finally:
        out.free()
        attr.close()
        if own:
            nbr.close()


def run_permutations(arg_tuple, **kwargs):
    """safepy/safe_extras.py:36-70.  arg_tuple = (neighborhood2node, node2attribute,
    neighborhood_score_type, num_permutations, random_seed); returns (counts_neg, counts_pos),
    float64 [N,M].  Like the reference it reseeds the *global* legacy NumPy RNG as a side
    effect (safe_extras.py:46); `verbose` is accepted and ignored (no progress bar)."""
    neighborhood2node, node2attribute, neighborhood_score_type, num_permutations, random_seed = arg_tuple
    device = kwargs.get('device', 0)
    np.random.seed(random_seed)            # the documented side effect; the stream itself runs in the library
    ctx = be.Context.default(device)
    nbr, own = _nbr_handle(ctx, neighborhood2node)
    attr = be.Attributes.from_host(ctx, node2attribute)
    perms = be.Permutations(ctx, attr.n, attr.row_flags(), int(num_permutations), random_seed)
    neg = ctx.alloc_f64(attr.n, attr.m)
    pos = ctx.alloc_f64(attr.n, attr.m)
    try:
        be.permtest_counts(ctx, nbr, attr, perms, neighborhood_score_type if neighborhood_score_type == 'z-score' else 'sum',
                           None, neg.ptr, pos.ptr)
        return neg.download((attr.n, attr.m)), pos.download((attr.n, attr.m))
    finally:
        neg.free()
        pos.free()
        perms.close()
        attr.close()
        if own:
            nbr.close()
