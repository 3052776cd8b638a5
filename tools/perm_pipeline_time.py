"""Pace of the permutation-table pipeline alone (host draws -> swaps -> upload -> scan), no enrichment kernel:
time from safe_perms_create to the last row of the table being on the device, at configs[1] (3789 movable rows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safepy_amd import backend as be
ctx = be.Context.default(0)
be.pin_threads_to_device_numa(0)
n = 3971
flags = np.ones(n, dtype=np.uint8); flags[:182] = 0
for P in (1000, 1000, 1000, 128, 256, 2000):
    ctx.sync()
    t = time.perf_counter()
    perms = be.Permutations(ctx, n, flags, P, 0)
    last = perms.read(P - 1, P)          # waits for the whole table, copies one row
    dt = time.perf_counter() - t
    print('P = %4d: %.2f ms  (%.3f ms per 128 permutations)' % (P, 1e3 * dt, 1e3 * dt / P * 128))
    perms.close()
