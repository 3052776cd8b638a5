"""Where the first ~0.2 ms of a seeded step go before the permutation handle exists (host wall clock, mean of 100 steps):
Attributes.from_device, attr.stats() (kernel + read-back), attr.row_flags(), Permutations(...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import safepy_amd
from safepy_amd import backend as be, workloads
be.pin_threads_to_device_numa(0)
torch.set_num_threads(1)
data = workloads.costanzo_surrogate(seed=0)
ctx = be.Context.default(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
b = data['attributes']; n, m = b.shape
b_dev = torch.from_numpy(np.ascontiguousarray(b.T)).to('cuda')
acc = np.zeros(5)
N = 100
for it in range(N + 10):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    attr = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F'); t.append(time.perf_counter())
    stats = attr.stats(); t.append(time.perf_counter())
    flags = attr.row_flags(); t.append(time.perf_counter())
    perms = be.Permutations(ctx, n, flags, 1000, 0); t.append(time.perf_counter())
    perms.read(0, 16); t.append(time.perf_counter())
    perms.close(); attr.close()
    if it >= 10:
        acc += np.diff(t)
print('from_device %.1f us | stats %.1f us | row_flags %.1f us | Permutations() %.1f us | first 16 rows on the device + read %.1f us' % tuple(1e6 * acc / N))
