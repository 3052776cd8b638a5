"""Read-back of a 139 MB result matrix (configs[1]: 3971 x 4373 f64) into different kinds of host memory."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np                                   # noqa: E402
from safepy_amd import backend as be                 # noqa: E402

print('THP enabled:', open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), '| defrag:', open('/sys/kernel/mm/transparent_hugepage/defrag').read().strip())
ctx = be.Context.default(0)
n, m = 3971, 4373
dev = ctx.alloc_f64(n, m)
libc = ctypes.CDLL('libc.so.6', use_errno=True)
MADV_HUGEPAGE = 14


def timed(label, make, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        a = make()
        t1 = time.perf_counter()
        dev.download_into(a) if hasattr(dev, 'download_into') else be.lib.safe_memcpy_d2h(ctx.handle, ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(dev.ptr), ctypes.c_size_t(a.nbytes))
        t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        del a
    print('%-34s alloc %.2f ms, copy %.2f ms (%.1f GB/s)' % (label, np.median([t[0] for t in ts]), np.median([t[1] for t in ts]), n * m * 8 / 1e6 / np.median([t[1] for t in ts])))


def fresh():
    return np.empty((n, m))


def fresh_huge():
    a = np.empty((n, m))
    addr = a.ctypes.data
    lo = (addr + (1 << 21) - 1) & ~((1 << 21) - 1)
    hi = (addr + a.nbytes) & ~((1 << 21) - 1)
    if hi > lo:
        libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), MADV_HUGEPAGE)
    return a


warm = np.empty((n, m))
warm[:] = 0
timed('fresh np.empty', fresh)
timed('fresh np.empty + MADV_HUGEPAGE', fresh_huge)
timed('reused (warm) array', lambda: warm)
for thr in ('1', '2', '8'):
    os.environ['SAFE_HIP_D2H_THREADS'] = thr
    timed('fresh, %s copy threads' % thr, fresh)
    timed('warm, %s copy threads' % thr, lambda: warm)
os.environ['SAFE_HIP_D2H_THREADS'] = '0'
timed('warm, plain hipMemcpy', lambda: warm)
import torch                                          # noqa: E402
pin = torch.empty((n, m), dtype=torch.float64).pin_memory().numpy()
timed('pinned (torch), plain hipMemcpy', lambda: pin)
