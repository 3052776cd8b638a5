"""BASELINE.json configs[3] / configs[4] at one rank's size on one MI355X (diagnostic, not the
bench line): N = 20 000 uniform layout, euclidean r = 0.1.
  hyper : M binary attributes through compute_pvalues 'auto' (hypergeometric path)
  quant : M quantitative f64 attributes x P permutations (one rank's share of config 5 is
          M = 6250, P = 1000; use a smaller P and scale linearly)
usage: bench_big.py hyper M | quant M P [score]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safepy_amd
from safepy_amd import backend as be, workloads

mode = sys.argv[1]
m = int(sys.argv[2])
n = int(os.environ.get('BIG_N', 20000))
xy = workloads.uniform_layout(4, n)
ctx = be.Context.default(0)
nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
t = time.perf_counter()
nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
ctx.sync()
print('define_neighborhoods (euclidean, device-resident): %.1f ms, nnz=%d (%.1f per node)' % (1e3 * (time.perf_counter() - t), nbr.nnz, nbr.nnz / n))
rng = np.random.default_rng(5)
if mode == 'hyper':
    b = (rng.uniform(size=(n, m)) < 0.01).astype(np.float32)
    attr = be.Attributes.from_host(ctx, b)
    outs = [ctx.alloc_f64(n, m) for _ in range(3)] + [ctx.alloc_f64(m)]
    calls, kern = [], []
    for it in range(int(os.environ.get('BIG_ITERS', 3))):
        ctx.sync(); t = time.perf_counter()
        be.hypergeom(ctx, nbr, attr, 0.05, [o.ptr for o in outs]); ctx.sync()
        dt = time.perf_counter() - t
        calls.append(1e3 * dt), kern.append(ctx.last_kernel()[1])
        if it < 3:
            print('hypergeom call %.2f ms -> %.3g enrichments/s; last kernel %s' % (1e3 * dt, n * m / dt, ctx.last_kernel()))
    if len(calls) > 3:
        print('%d calls: call min %.3f median %.3f ms; %s min %.3f median %.3f ms' % (len(calls), min(calls[1:]), float(np.median(calls[1:])),
              ctx.last_kernel()[0], min(kern[1:]), float(np.median(kern[1:]))))
else:
    P = int(sys.argv[3])
    score = sys.argv[4] if len(sys.argv) > 4 else 'sum'
    dtype = np.float32 if os.environ.get('BIG_F32') else np.float64
    b = workloads.quantitative_attributes(3, n, m, dtype=dtype)
    attr = be.Attributes.from_host(ctx, b)
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    for it in range(2):
        perms = be.Permutations(ctx, n, attr.row_flags(), P, 0)
        ctx.sync(); t = time.perf_counter()
        be.randomization(ctx, nbr, attr, perms, score, 'both', 0.05, [o.ptr for o in outs]); ctx.sync()
        dt = time.perf_counter() - t
        perms.close()
        name, kms, kl = ctx.last_kernel()
        core, undecided = be.last_mfma_filter(ctx)
        print('%s %s: call %.1f ms, kernel %.2f ms x %d -> %.3g enrichments/s (n=%d m=%d P=%d); config-5 rank share (6250 x 1000) ~ %.2f s; %d slices on the matrix cores, %d compares (%.2e) settled exactly'
              % (score, name, 1e3 * dt, kms, kl, n * m * P / dt, n, m, P, dt * (6250 / m) * (1000 / P), core, undecided, undecided / (float(n) * m * P)))
