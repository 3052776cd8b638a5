"""Times the two exchange steps of the sharded path on ONE rank (RCCL group of size 1): the
fixed software cost of the collectives, without any link traffic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import safepy_amd
from safepy_amd import backend as be, workloads, sharding

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0); torch.set_num_threads(int(os.environ.get("TORCH_THREADS", 1)))
print('affinity before init:', len(os.sched_getaffinity(0)), 'cpus')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
_t = torch.zeros(4, device='cuda'); dist.all_reduce(_t); torch.cuda.synchronize()
print('affinity after init + first collective:', len(os.sched_getaffinity(0)), 'cpus', sorted(os.sched_getaffinity(0))[:8])
if os.environ.get('RESET_AFFINITY'):
    os.sched_setaffinity(0, range(os.cpu_count()))
    print('affinity reset:', len(os.sched_getaffinity(0)))
data = workloads.costanzo_surrogate(seed=0)
ctx = be.Context.default(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
nbr = sf._nbr
b = data['attributes']; n, m = b.shape
b_dev = torch.from_numpy(np.ascontiguousarray(b.T)).to('cuda')
out = [torch.empty((n, m), dtype=torch.float64, device='cuda') for _ in range(5)]
enr = torch.empty(m, dtype=torch.float64, device='cuda')
P = 1000
table = be.nes_table(P)
def sync():
    torch.cuda.synchronize()
import gc
if os.environ.get('GC_FREEZE'):
    gc.collect(); gc.freeze()
if os.environ.get('GC_DEBUG'):
    gc.callbacks.append(lambda phase, info: print('gc', phase, info) if info.get('generation') == 2 else None)
for it in range(int(os.environ.get("ITERS", 12))):
    T = []
    sync(); t0 = time.perf_counter(); c0 = time.process_time()
    attr = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F')
    stats = attr.stats(); flags = attr.row_flags(); sync(); T.append(('stats+flags', time.perf_counter()))
    if not os.environ.get('NOCOLL'):
        flags, stats = sharding.reduce_flags_and_stats(flags, stats)
    sync(); T.append(('reduce_flags_and_stats', time.perf_counter()))
    attr.set_row_flags(flags); T.append(('set_row_flags', time.perf_counter()))
    perms = be.Permutations(ctx, n, flags, P, 0); T.append(('perms_create', time.perf_counter()))
    be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [t.data_ptr() for t in out] + [enr.data_ptr()], table=table); sync(); T.append(('randomization', time.perf_counter()))
    if not os.environ.get('NOCOLL'):
        full = sharding.gather_nes(ctx, nbr, out[3], m, P, 'both', table=table); sync(); T.append(('gather_nes', time.perf_counter()))
        if not os.environ.get('NOF64'):
            old = sharding.gather_columns(out[3], m); sync(); T.append(('gather_columns(f64)', time.perf_counter()))
        assert torch.equal(full, out[3])
    perms.close(); attr.close()
    prev = t0; line = []
    for k, v in T:
        line.append('%s %.2f' % (k, 1e3 * (v - prev))); prev = v
    print('iter', it, 'cpu %.1f ms' % (1e3 * (time.process_time() - c0)), ' | '.join(line))
    if os.environ.get('SLEEP_MS'): time.sleep(float(os.environ['SLEEP_MS']) / 1e3)
dist.destroy_process_group()
