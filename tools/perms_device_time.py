"""Time of the device-side permutation generator (safe_perms_create_device) against the host stream: tables of P permutations
of the configs[1] row set, complete on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from safepy_amd import backend as be

ctx = be.Context.default(0)
for n, k_fixed in ((3971, 182), (20000, 1000)):
    flags = np.ones(n, dtype=np.uint8)
    flags[np.random.default_rng(0).choice(n, k_fixed, replace=False)] = 0
    for P in (1000, 10000):
        for mode in ('device', 'host'):
            best = 1e9
            for it in range(4):
                ctx.sync()
                t0 = time.perf_counter()
                perms = be.Permutations(ctx, n, flags, P, None if mode == 'device' else 1, device_key=5)
                perms.read(P - 1, P)                     # the last row: everything before it is complete
                dt = time.perf_counter() - t0
                perms.close()
                if it:
                    best = min(best, dt)
            print('n=%d P=%d %s stream: tables complete after %.2f ms' % (n, P, mode, 1e3 * best), flush=True)
