"""mfma vs lds f64 permutation kernels as a function of the number of quantitative columns (the dispatch rule of enrich.hip)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    import safepy_amd
    from safepy_amd import backend as be, workloads
    n, P = int(sys.argv[1]), int(sys.argv[2])
    ctx = be.Context.default(0)
    xy = workloads.clustered_layout(np.random.default_rng(1), n)
    nbr = be.Neighborhoods.euclidean(ctx, xy, 0.06 * np.ptp(xy[:, 0]))
    for m in (1, 4, 16, 32, 64, 128, 256, 512, 1024):
        b = np.round(np.random.default_rng(m).normal(size=(n, m)) * 1024) / 1024
        attr = be.Attributes.from_host(ctx, b)
        bufs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
        ts = []
        for rep in range(4):
            perms = be.Permutations(ctx, n, attr.row_flags(), P, None, device_key=5)
            t0 = time.perf_counter()
            be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [x.ptr for x in bufs])
            ctx.sync()
            ts.append(1e3 * (time.perf_counter() - t0))
            perms.close()
        print('%s n=%d P=%d m=%4d: %.2f ms (%s)' % (os.environ.get('SAFE_HIP_FORCE_PATH', 'default'), n, P, m, min(ts), ctx.last_kernel()[0]), flush=True)
        for x in bufs: x.free()
        attr.close()
else:
    for n, P in ((1586, 2000), (3971, 1000), (20000, 256)):
        for path in ('mfma', 'lds'):
            subprocess.run([sys.executable, __file__, str(n), str(P)], env=dict(os.environ, SAFE_HIP_FORCE_PATH=path))
