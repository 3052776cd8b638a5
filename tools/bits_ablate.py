"""Kernel-only time of the bit-sliced permutation test at configs[1] (tables generated before the call), per kernel
variant and diagnostic build:  SAFE_HIP_BITS_KERNEL = blk | pre;  SAFE_HIP_BITS_DBG bit 0 = no LDS gathers, bit 1 = no
counter flush, bit 2 = no compare / count (make DIAG=1: wrong results; shows what the time goes to), 128 = the per-task
clock trace (results stay correct).
usage: bits_ablate.py [P]   -- each configuration runs in its own child process."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(P):
    import numpy as np
    import safepy_amd
    from safepy_amd import backend as be, workloads
    be.pin_threads_to_device_numa(0)
    data = workloads.costanzo_surrogate(seed=0)
    ctx = be.Context.default(0)
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
    sf.define_neighborhoods()
    nbr = sf._nbr
    b = data['attributes']; n, m = b.shape
    if os.environ.get('SORT_COLS') == '1':            # experiment: attributes in order of annotation count (dense ones share word groups)
        b = np.asfortranarray(b[:, np.argsort(np.nansum(b, axis=0), kind='stable')])
    attr = be.Attributes.from_host(ctx, b)
    attr.stats()
    flags = attr.row_flags()
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    table = be.nes_table(P)
    best = None
    for it in range(5):
        perms = be.Permutations(ctx, n, flags, P, 0)
        perms.read(P - 1, P)                      # the whole table is on the device now
        ctx.sync(); t0 = time.perf_counter()
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs], table=table)
        ctx.sync(); dt = time.perf_counter() - t0
        name, ms, launches = ctx.last_kernel()
        perms.close()
        if it and (best is None or dt < best[0]):
            best = (dt, name, ms, launches)
    dt, name, ms, launches = best
    print('%-22s dbg=%s: call %.2f ms, %d launches x %.3f ms = %.2f ms' % (
        name, os.environ.get('SAFE_HIP_BITS_DBG', '0'), 1e3 * dt, launches, ms, ms * launches), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--one':
        one(int(sys.argv[2]))
        sys.exit(0)
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    K = 'SAFE_HIP_BITS_KERNEL'
    D = 'SAFE_HIP_BITS_DBG'
    configs = [{K: 'pre'}, {K: 'blk'}, {K: 'blk', D: '1'}, {K: 'blk', D: '2'}, {K: 'blk', D: '4'}, {K: 'blk', D: '7'}]
    for cfg in configs:
        subprocess.run([sys.executable, os.path.abspath(__file__), '--one', str(P)], env=dict(os.environ, **cfg))
