import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
SEED = None if os.environ.get('TRACE_SEED', '0') == 'none' else int(os.environ.get('TRACE_SEED', '0'))   # 'none': unseeded (device stream)
def step():
    attr = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F')
    sharding.randomization_step(ctx, nbr, attr, m, 1000, SEED, out, enr, table=table)
    attr.close()
for _ in range(3): step()
torch.cuda.synchronize()
sys.stderr.write('==== traced step\n')
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print('step ms', 1e3 * (time.perf_counter() - t0))
