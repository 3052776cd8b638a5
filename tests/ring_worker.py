"""One rank of the node-shared permutation stream's protocol test (tests/test_ring_cpu.py): no GPU, only the ring
entry points of the C ABI.  argv: name local_rank local_world."""
import ctypes as C
import sys

import numpy as np


def pattern(call, chunk, words):
    return (np.arange(words, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(call * 1000003 + chunk * 7919)).astype(np.uint32)


def main():
    name, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from safepy_amd import _lib
    lib, check = _lib.lib, _lib.check
    ring = C.c_void_p()
    slot = 64 * 1024
    check(lib.safe_ring_open(name.encode(), rank, world, 4 * slot + 100, C.byref(ring)))       # room for 4 slots
    words = slot // 4
    # call 1: 20 chunks through 4 slots (the producer must wait for the slowest consumer); local rank 2 leaves after 3 chunks
    # call 2: another shape, ragged last chunk; call 3: local rank 1 asks for another permutation count -> SAFE_E_VALUE there,
    # the others complete
    for call, (n, count, chunks) in enumerate([(1000, 2560, 20), (777, 300, 3), (50, 128, 1)], start=1):
        my_count = count + (1 if (call == 3 and rank == 1) else 0)
        rc = lib.safe_ring_begin(ring, n, n - 5, my_count, 0xABCDEF0123456789, slot)
        if call == 3 and rank == 1:
            assert rc == _lib.E_VALUE, rc
            assert b'differs from' in lib.safe_last_error()
            continue
        check(rc)
        for ci in range(chunks):
            nbytes = slot if ci + 1 < chunks else slot // 2 + 4
            if rank == 0:
                src = pattern(call, ci, words)
                check(lib.safe_ring_publish(ring, ci, src.ctypes.data, nbytes))
            else:
                if call == 1 and rank == 2 and ci == 3:
                    break                                            # leaves early: must not hold the producer up
                dst = np.zeros(words, dtype=np.uint32)
                check(lib.safe_ring_fetch(ring, ci, dst.ctypes.data, nbytes))
                assert np.array_equal(dst[:nbytes // 4], pattern(call, ci, words)[:nbytes // 4]), (call, ci)
                assert not dst[nbytes // 4:].any()
        check(lib.safe_ring_end(ring))
    if rank != 0:                                                    # out-of-order fetches are refused
        check(lib.safe_ring_begin(ring, 10, 5, 128, 1, slot))
        dst = np.zeros(words, dtype=np.uint32)
        assert lib.safe_ring_fetch(ring, 1, dst.ctypes.data, 16) == _lib.E_INVALID
        check(lib.safe_ring_fetch(ring, 0, dst.ctypes.data, 16))
        check(lib.safe_ring_end(ring))
    else:
        check(lib.safe_ring_begin(ring, 10, 5, 128, 1, slot))
        check(lib.safe_ring_publish(ring, 0, pattern(9, 0, words).ctypes.data, 16))
        check(lib.safe_ring_end(ring))
    check(lib.safe_ring_close(ring))
    print('ring rank %d ok' % rank)


if __name__ == '__main__':
    main()
