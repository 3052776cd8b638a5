"""Device-to-host copy of a result matrix into a fresh NumPy array (backend.DeviceBuffer.download = safe_memcpy_d2h): the
plain pageable copy vs the pipelined form with k copy threads (SAFE_HIP_D2H_THREADS)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    import torch
    import safepy_amd
    from safepy_amd import backend as be
    ctx = be.Context.default(0)
    for n, m in ((3971, 4373), (20000, 10000), (1000, 8400), (517, 16397)):
        src = torch.arange(n * m, dtype=torch.float64, device='cuda').reshape(n, m) * 0.5
        torch.cuda.synchronize()
        buf = be.DeviceBuffer.__new__(be.DeviceBuffer)
        buf.ctx, buf.nbytes, buf.ptr = ctx, n * m * 8, src.data_ptr()
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            out = buf.download((n, m))
            ts.append(1e3 * (time.perf_counter() - t0))
            ok = bool(out[0, 0] == 0.0 and out[-1, -1] == (n * m - 1) * 0.5 and out[n // 2, m // 3] == ((n // 2) * m + m // 3) * 0.5)
            if rep == 0:
                ok = ok and np.array_equal(out, src.cpu().numpy())
            del out
        buf.ptr = None
        print('threads=%s  %d x %d (%.0f MB): %.2f ms best, %.2f median -> %.1f GB/s  %s' % (
            os.environ.get('SAFE_HIP_D2H_THREADS', 'default'), n, m, n * m * 8 / 1e6, min(ts), sorted(ts)[len(ts) // 2], n * m * 8 / min(ts) / 1e6, 'ok' if ok else 'WRONG'), flush=True)
else:
    for k in ('0', '2', '4', '6', '8', '12'):
        subprocess.run([sys.executable, __file__, 'x'], env=dict(os.environ, SAFE_HIP_D2H_THREADS=k))
