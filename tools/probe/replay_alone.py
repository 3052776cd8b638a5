"""The seeded permutation stream alone (no enrichment kernels): rocprofv3 --kernel-trace --stats of this script gives the
durations of k_replay_targets / k_scan_round / k_emit_rows without contention."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import safepy_amd
from safepy_amd import backend as be
ctx = safepy_amd.Context.default(0)
n = 3971
flags = np.ones(n, dtype=np.uint8); flags[np.random.default_rng(0).choice(n, 182, replace=False)] = 0
for rep in range(5):
    t0 = time.perf_counter()
    p = be.Permutations(ctx, n, flags, int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 0)
    p.read(0, 1)
    t1 = time.perf_counter()
    x = p.read()
    t2 = time.perf_counter()
    tm = p.timing()
    p.close()
    print('first row after %.3f ms, all rows read after %.3f ms; draw_busy %.3f drawn_all %.3f enqueued_all %.3f' % (1e3 * (t1 - t0), 1e3 * (t2 - t0), tm['draw_busy_ms'], tm['drawn_all_ms'], tm['tables_enqueued_ms']))
