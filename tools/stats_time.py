import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from safepy_amd import backend as be, workloads
ctx = be.Context.default(0)
n, m = 3971, 4373
rng = np.random.default_rng(1)
for name, b in (('binary f32 F', np.asfortranarray((rng.uniform(size=(n, m)) < 0.01).astype(np.float32))),
                ('quantitative f64 C', rng.normal(size=(n, m))),
                ('quantitative f32 F', np.asfortranarray(rng.normal(size=(n, m)).astype(np.float32)))):
    ts = []
    for it in range(6):
        attr = be.Attributes.from_host(ctx, b)
        ctx.sync(); t = time.perf_counter(); st = attr.stats(); ts.append(1e3 * (time.perf_counter() - t)); attr.close()
    print('%-20s stats() %.3f ms (min of 6)  %s' % (name, min(ts), st))
