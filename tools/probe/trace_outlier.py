"""Find what a slow headline step was waiting for: runs N seeded steps with SAFE_HIP_TRACE=1 (host-side event log of the
library on stderr, redirected to a file by the caller), marks every step, and afterwards prints, for the slowest steps, the
trace lines that are followed by a gap of more than 0.25 ms.  usage: SAFE_HIP_TRACE=1 python trace_outlier.py N 2> trace.log; then
python trace_outlier.py --parse trace.log"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    steps, cur = [], None
    for line in open(sys.argv[2], errors='replace'):
        if line.startswith('==== step'):
            if cur is not None:
                steps.append(cur)
            cur = {'id': int(line.split()[2]), 'lines': [], 'ms': None}
        elif line.startswith('==== done') and cur is not None:
            cur['ms'] = float(line.split()[2])
        elif cur is not None:
            m = re.match(r'\[safe_hip\s+([0-9.]+) ms\] (.*)', line)
            if m:
                cur['lines'].append((float(m.group(1)), m.group(2)))
    if cur is not None:
        steps.append(cur)
    steps = [s for s in steps if s['ms'] is not None]
    ms = sorted(s['ms'] for s in steps)
    med = ms[len(ms) // 2]
    print('%d steps, median %.3f ms, mean %.3f, max %.3f; %d steps > 1.3 x median, %d > 2 x median' % (
        len(steps), med, sum(ms) / len(ms), ms[-1], sum(1 for v in ms if v > 1.3 * med), sum(1 for v in ms if v > 2 * med)))
    for s in sorted(steps, key=lambda s: -s['ms'])[:4]:
        print('--- step %d: %.3f ms' % (s['id'], s['ms']))
        ls = s['lines']
        for (t0, a), (t1, b) in zip(ls, ls[1:]):
            if t1 - t0 > 0.25:
                print('    %.3f ms between "%s" and "%s"' % (t1 - t0, a.strip(), b.strip()))
    if len(sys.argv) > 3:                                         # full trace of the slowest step and of a typical one
        for s in (max(steps, key=lambda s: s['ms']), sorted(steps, key=lambda s: s['ms'])[len(steps) // 2]):
            print('=== full trace of step %d (%.3f ms)' % (s['id'], s['ms']))
            t_first = s['lines'][0][0] if s['lines'] else 0.0
            for t, what in s['lines']:
                print('  %8.3f  %s' % (t - t_first, what.rstrip()))
    sys.exit(0)

import numpy as np, torch
import safepy_amd
from safepy_amd import backend as be, workloads, sharding
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
def step():
    attr = be.Attributes.from_device(ctx, b_dev.data_ptr(), np.float32, n, m, order='F')
    sharding.randomization_step(ctx, nbr, attr, m, 1000, 0, out, enr, table=table)
    attr.close()
for _ in range(3): step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze(); gc.disable()
for i in range(N):
    sys.stderr.write('==== step %d\n' % i); sys.stderr.flush()
    t0 = time.perf_counter(); step(); dt = 1e3 * (time.perf_counter() - t0)
    sys.stderr.write('==== done %.3f\n' % dt); sys.stderr.flush()
