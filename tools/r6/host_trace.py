"""Host-side timeline of one seeded headline step: SAFE_HIP_TRACE=1 lines of the library + Python-level stamps around the calls of
sharding.randomization_step (what the host does between the step's start and the first permutation kernel)."""
import os, sys, time
os.environ['SAFE_HIP_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import safepy_amd
from safepy_amd import backend as be, workloads, sharding
be.pin_threads_to_device_numa(0)
torch.set_num_threads(1)
data = workloads.costanzo_surrogate(seed=0)
ctx = be.Context.default(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
nbr = sf._nbr
b = data['attributes']; n, m = b.shape
b_dev = torch.from_numpy(np.ascontiguousarray(b.T)).to('cuda')
out = {k: torch.empty((n, m), dtype=torch.float64, device='cuda') for k in sharding.RANDOMIZATION_OUTPUTS}
enr = torch.empty((m,), dtype=torch.float64, device='cuda')
table = be.nes_table(1000)
def step(stamp=False):
    t0 = time.perf_counter()
    attr = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F')
    t1 = time.perf_counter()
    stats = attr.stats(); t2 = time.perf_counter()
    flags = attr.row_flags(); t3 = time.perf_counter()
    perms = be.Permutations(ctx, n, flags, 1000, 0); t4 = time.perf_counter()
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [out[k].data_ptr() for k in sharding.RANDOMIZATION_OUTPUTS] + [enr.data_ptr()], table=table)
    t5 = time.perf_counter()
    perms.close(); attr.close(); t6 = time.perf_counter()
    if stamp:
        sys.stderr.write('PY attr %.1f | stats %.1f | row_flags %.1f | Permutations %.1f | randomization %.1f | close %.1f us; total %.1f\n'
                         % tuple(1e6 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)))
for _ in range(5): step()
torch.cuda.synchronize()
sys.stderr.write('==== traced step\n')
step(True)
sys.stderr.write('==== traced step 2\n')
step(True)
