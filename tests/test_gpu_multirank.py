"""N > 1 on real devices: `world` fresh child processes run the product's sharded driver and check
sharded == unsharded == oracle (tests/multirank_worker.py).  Two forms:

  * RCCL: one GPU per rank over the nccl backend -- needs >= 2 visible devices, skipped otherwise;
  * one-GPU form: two ranks share device 0 and exchange through gloo (host-staged) -- the same kernels,
    the same integer-counter exchange and seed agreement, runnable on the single-GPU test box.

`bench.py --gpus N` without a launcher must start its own ranks: checked with N = 2 when two devices exist,
and for every box through the launcher-less single-rank RCCL group (SAFE_BENCH_FORCE_DIST)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _device_count():
    import safepy_amd
    return safepy_amd.device_count()


def _run_ranks(world, backend, tmp_path, timeout=900, stream='shared'):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1', GPU_MAX_HW_QUEUES='8',
               SAFE_HIP_RING_TIMEOUT_S='300')
    # every rank writes to its own file: a rank blocked on a full pipe inside a collective would stall all of them
    logs = [open(tmp_path / ('rank%d.log' % r), 'w+') for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'multirank_worker.py'), str(r), str(world), str(port),
                               backend, str(tmp_path), stream], env=env, stdout=logs[r], stderr=subprocess.STDOUT)
             for r in range(world)]
    timed_out = False
    for p in procs:
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            timed_out = True
            for q in procs:
                q.kill()
    text = []
    for log in logs:
        log.seek(0)
        text.append(log.read())
        log.close()
    assert not timed_out, 'TIMEOUT\n' + '\n'.join('--- rank %d:\n%s' % (r, t[-3000:]) for r, t in enumerate(text))
    for r, p in enumerate(procs):
        assert p.returncode == 0 and (tmp_path / ('ok%d' % r)).exists(), 'rank %d failed:\n%s' % (r, text[r][-4000:])


@pytest.mark.parametrize('stream', ['shared', 'own'])
def test_two_ranks_one_gpu_sharded_equals_unsharded_equals_oracle(tmp_path, stream):
    if _device_count() < 1:
        pytest.skip('needs a HIP device')
    _run_ranks(2, 'gloo', tmp_path, stream=stream)


def test_three_ranks_one_gpu_uneven_split(tmp_path):
    if _device_count() < 1:
        pytest.skip('needs a HIP device')
    _run_ranks(3, 'gloo', tmp_path)


def test_eight_ranks_one_gpu_seven_consumers(tmp_path):
    """The rank count of the driver's scaling run: one producer of the node's permutation stream, SEVEN consumers on the ring,
    columns split eight ways (np.array_split sizes), sharded == unsharded == oracle on every rank."""
    if _device_count() < 1:
        pytest.skip('needs a HIP device')
    _run_ranks(8, 'gloo', tmp_path, timeout=1200)


def test_two_ranks_rccl_sharded_equals_unsharded_equals_oracle(tmp_path):
    if _device_count() < 2:
        pytest.skip('RCCL with 2 ranks needs 2 devices (one GPU per rank)')
    _run_ranks(2, 'nccl', tmp_path)


def _bench_line(args, env=None, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + args
    res = subprocess.run(cmd, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=timeout, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) prints one JSON line with n_gpus 2."""
    if _device_count() < 2:
        pytest.skip('needs 2 devices')
    line = _bench_line(['--gpus', '2', '--steps', '3', '--warmup', '1', '--nodes', '1200', '--attrs', '640', '--perms', '200',
                        '--cpu-perms', '0', '--extras', '0'])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['exchange']['d2h_only_ms_per_step'] > 0 and line['exchange']['all_gather_ms_per_step'] > 0


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """What the driver's `bench.py --gpus 8` does, rehearsed on one GPU (SAFE_BENCH_SHARE_DEVICE=1: every rank on device 0, the
    exchange staged through gloo) under the bench hosts' 16-CPU quota: the launcher-less start, eight ranks, the shared stream
    with seven consumers, and the multi_gpu_configs extras (configs[2] strong scaling seeded and unseeded, one configs[4] rank
    share) run to the end well inside ten minutes and the line is complete.  (The GPU is time-shared: `value` means nothing.)"""
    if _device_count() < 1:
        pytest.skip('needs a HIP device')
    import shutil
    import time
    pre = ['taskset', '-c', '0-15'] if shutil.which('taskset') and (os.cpu_count() or 1) >= 16 else []
    t0 = time.time()
    res = subprocess.run(pre + [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '5', '--warmup', '1', '--cpu-perms', '0'],
                         env=dict(os.environ, SAFE_BENCH_SHARE_DEVICE='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, cwd=ROOT)
    wall = time.time() - t0
    assert res.returncode == 0, res.stderr[-4000:]
    assert 'not published' not in res.stderr and 'Traceback' not in res.stderr, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert wall < 600, wall
    assert line['n_gpus'] == 8 and len(line['per_rank']) == 8
    roles = [r['role'] for r in line['per_rank']]
    assert roles.count('producer') == 1 and roles.count('consumer') == 7, roles
    extras = line['multi_gpu_configs']
    assert set(extras) == {'configs2_strong_scaling', 'configs2_strong_scaling_unseeded', 'configs4_rank_share'}
    assert all(r['role'] == 'device' for r in extras['configs2_strong_scaling_unseeded']['per_rank'])
    assert extras['configs2_strong_scaling']['attributes_per_gpu'] in (546, 547)
    # the producer rank's host CPU per step stays within 1.3 x the step (sleeping waits: 2 cores per rank)
    x = line['exchange']
    assert max(x['host_cpu_ms_per_step_no_exchange_per_rank']) <= 1.3 * x['no_exchange_ms_per_step'], x


def test_bench_single_rank_rccl_group_reports_the_exchange():
    """The N > 1 code path of bench.py (process group, flag/seed exchange, all-gather, the D2H-only comparison)
    on a one-rank RCCL group."""
    if _device_count() < 1:
        pytest.skip('needs a HIP device')
    line = _bench_line(['--gpus', '1', '--steps', '3', '--warmup', '1', '--nodes', '1200', '--attrs', '640', '--perms', '200',
                        '--cpu-perms', '0', '--extras', '0'],
                       env={'SAFE_BENCH_FORCE_DIST': '1', 'MASTER_PORT': str(_free_port())})
    assert line['n_gpus'] == 1 and 'exchange' in line and line['exchange']['form'].startswith('packed 10 + 10 bit')
