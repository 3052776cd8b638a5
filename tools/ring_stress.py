#!/usr/bin/env python3
"""Stress of the node-shared permutation stream on ONE GPU: WORLD ranks (gloo) make ITERS collective Permutations(shared=True)
calls of random sizes with random per-rank delays and partial consumption, and compare table checksums.
usage: ring_stress.py WORLD ITERS            (parent)   |   ring_stress.py RANK WORLD PORT ITERS   (rank)"""
import os
import socket
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, world, port, iters):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from safepy_amd import backend as be, sharding
    ctx = be.Context.default(0)
    assert sharding.ensure_shared_stream(ctx)
    plan = np.random.default_rng(7)              # the same plan on every rank
    me = np.random.default_rng(100 + rank)       # this rank's own delays
    t0 = time.time()
    for it in range(iters):
        n = int(plan.choice([50, 700, 3971, 9000]))
        P = int(plan.choice([1, 33, 129, 300, 1000, 2500]))
        flags = (plan.uniform(size=n) < 0.95).astype(np.uint8)
        upto = int(plan.integers(1, P + 1))      # how much of the stream the ranks look at
        if me.uniform() < 0.3:
            time.sleep(me.uniform() * 0.02)
        perms = be.Permutations(ctx, n, flags, P, 11 + it, shared=True)
        role = perms.timing()['role']
        assert role == ('producer' if rank == 0 else 'consumer'), role
        partial = me.uniform() < 0.4
        rows = perms.read(0, upto if partial else P)
        if me.uniform() < 0.3:
            time.sleep(me.uniform() * 0.01)
        perms.close()
        crc = zlib.crc32(rows[:upto].tobytes())
        got = [None] * world
        dist.all_gather_object(got, crc)
        assert len(set(got)) == 1, (it, n, P, got)
        if it % 10 == 0 and rank == 0:
            own = be.Permutations(ctx, n, flags, P, 11 + it)
            assert zlib.crc32(own.read(0, upto).tobytes()) == crc, (it, n, P)
            own.close()
            print('iter %d ok (n=%d P=%d) %.1f s' % (it, n, P, time.time() - t0), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print('rank %d done' % rank, flush=True)


def main():
    if len(sys.argv) == 3:
        world, iters = int(sys.argv[1]), int(sys.argv[2])
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = dict(os.environ, OMP_NUM_THREADS='1', SAFE_HIP_RING_TIMEOUT_S='60')
        procs = [subprocess.Popen([sys.executable, __file__, str(r), str(world), str(port), str(iters)], env=env) for r in range(world)]
        rcs = [p.wait() for p in procs]
        print('exit codes', rcs)
        sys.exit(max(rcs))
    rank_main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))


if __name__ == '__main__':
    main()
