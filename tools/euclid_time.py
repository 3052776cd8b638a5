"""k_euclid_dense (all-pairs distance + threshold, int64 [N,N] out) timed with HIP events; usage: euclid_time.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safepy_amd import backend as be
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = be.Context.default(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
xy = np.random.default_rng(4).uniform(size=(n, 2))
t_xy = torch.from_numpy(xy).to('cuda')
t_mask = torch.empty((n, n), dtype=torch.int64, device='cuda')
nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
for _ in range(3):
    ctx.euclidean_dense(t_xy.data_ptr(), n, nr, t_mask.data_ptr(), None)
for rep in range(3):
    ctx.timer_start()
    for _ in range(10):
        ctx.euclidean_dense(t_xy.data_ptr(), n, nr, t_mask.data_ptr(), None)
    ms = ctx.timer_stop_ms() / 10
    print('N = %d: %.3f ms, %.2f TB/s' % (n, ms, (16 * n + 8 * n * n) / ms / 1e9))
