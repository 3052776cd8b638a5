"""The node-shared permutation stream's shared-memory protocol (safepy_amd/csrc/ring.cpp) between real processes, on
the CPU: three local ranks, several calls in a row, more chunks than slots, a consumer that leaves early, a rank whose call
disagrees, ordered fetches.  The GPU side of the same path is tests/test_gpu_multirank.py."""
import glob
import os
import subprocess
import sys
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, name):
    env = dict(os.environ, PYTHONPATH=ROOT, SAFE_HIP_RING_TIMEOUT_S='30')
    logs = []
    procs = []
    for r in range(world):
        log = open('/tmp/ring_%s_%d.log' % (name, r), 'w+')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'ring_worker.py'), name, str(r), str(world)],
                                      env=env, stdout=log, stderr=subprocess.STDOUT))
    out = []
    for p, log in zip(procs, logs):
        rc = p.wait(timeout=120)
        log.seek(0)
        out.append((rc, log.read()))
        log.close()
        os.unlink(log.name)
    return out


def test_three_local_ranks_share_one_stream():
    name = 'test' + uuid.uuid4().hex[:12]
    out = _run(3, name)
    for r, (rc, text) in enumerate(out):
        assert rc == 0 and ('ring rank %d ok' % r) in text, 'rank %d:\n%s' % (r, text)
    assert not glob.glob('/dev/shm/safe_hip.' + name + '*')          # the segment's name is gone once everyone has attached


def test_consumer_without_a_producer_times_out_with_a_message():
    import ctypes as C
    from safepy_amd import _lib
    ring = C.c_void_p()
    os.environ['SAFE_HIP_RING_TIMEOUT_S'] = '0.3'
    try:
        rc = _lib.lib.safe_ring_open(('absent' + uuid.uuid4().hex[:8]).encode(), 1, 2, 1 << 20, C.byref(ring))
    finally:
        del os.environ['SAFE_HIP_RING_TIMEOUT_S']
    assert rc == _lib.E_VALUE and b'did not appear' in _lib.lib.safe_last_error()
