import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, safepy_amd
from safepy_amd import backend as be
ctx = be.Context.default(0)
uid = be.Comm.unique_id()
comm = be.Comm(ctx, 1, 0, uid)
if len(sys.argv) > 1 and sys.argv[1] == 'gather':
    src, dst = ctx.alloc(1 << 20), ctx.alloc(1 << 20)
    src.upload(np.arange(1 << 20, dtype=np.uint8))
    dst.zero()
    comm.allgather(src.ptr, 1 << 20, dst.ptr)
    ctx.sync()
    src.free(); dst.free()
comm.close()
print('done', sys.argv[1:])
