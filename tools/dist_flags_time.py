"""Where the per-step cost of a process group goes (one-rank RCCL group on one GPU): the headline step without a group, with the
group initialised but the small exchange skipped (flags handed in), and with the flags / statistics / seed all-gather; plus the
pieces of that exchange.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29544', RANK='0', WORLD_SIZE='1')
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
torch.set_num_threads(1)
import safepy_amd
from safepy_amd import backend as be, workloads, sharding
be.pin_threads_to_device_numa(0)
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


def step(flags=None, exchange=False):
    attr = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F')
    if flags is not None:
        attr.stats()
    sharding.randomization_step(ctx, nbr, attr, m, 1000, 0, out, enr, table=table, flags=flags, exchange=exchange)
    attr.close()


def t(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps


print('step, no process group                 %.3f ms' % t(step))
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
attr0 = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F')
st, fl = attr0.stats(), attr0.row_flags()
print('step, group up, flags handed in        %.3f ms' % t(lambda: step(flags=fl)))
print('step, group up, small all-gather       %.3f ms' % t(step))
print('step, group up, + result all-gather    %.3f ms' % t(lambda: step(exchange=True)))
print('reduce_flags_and_stats alone           %.3f ms' % t(lambda: sharding.reduce_flags_and_stats(fl, st, None, 0)))
dist.destroy_process_group()
