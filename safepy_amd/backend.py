"""Thin object layer over the C ABI: device context, device buffers and the three
device-resident handles (membership, attribute matrix, permutation tables).

Nothing here computes: every method forwards to libsafe_hip.so, and fails loudly if the
library reports an error (no HIP device, bad arguments, ...).
"""
import collections
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import lib, check

_SCORE = {'sum': _lib.SCORE_SUM, 'z-score': _lib.SCORE_ZSCORE}
_SIGN = {'highest': _lib.SIGN_HIGHEST, 'lowest': _lib.SIGN_LOWEST, 'both': _lib.SIGN_BOTH}


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def _split_off_cores(cpus, n_cores, slot=0):
    """(CPUs of `n_cores` whole physical cores, the rest) of the CPU set `cpus`, or None when the topology cannot be read or
    fewer than n_cores + 2 cores would be left."""
    by_core = {}
    for c in sorted(cpus):
        try:
            sib = set()
            for part in open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c).read().strip().split(','):
                lo, _, hi = part.partition('-')
                sib.update(range(int(lo), int(hi or lo) + 1))
        except OSError:
            return None
        if sib <= cpus:                                   # only cores whose hardware threads are all ours to use
            by_core[min(sib)] = sib
    cores = sorted(by_core)
    if len(cores) < n_cores * (slot + 1) + 2:
        return None
    taken = []                                            # core after core (the library gives each of its two draw threads one half)
    hi = len(cores) - n_cores * slot                      # (ranks of one node take different cores: slot = local rank)
    for c in cores[hi - n_cores:hi]:
        taken += sorted(by_core[c])
    return taken, set(cpus) - set(taken)


def pin_threads_to_device_numa(device=0, reserve_draw_cores=0):
    """Keep every thread this process has -- and those it starts later: the draw thread -- on the CPUs of the NUMA node the GPU hangs off.  A container may be scheduled on any CPU of a
    two-socket host; with the draw thread on the far socket a whole run is ~15-20 % slower (5.2 vs 5.4-6.2
    ms/step at configs[1]).  Returns the node, or None when the topology cannot be read (nothing is changed
    then).  For launchers (bench.py, run_batch): a library does not re-pin its caller behind its back.
    reserve_draw_cores = k > 0: k whole physical cores of that node are set aside for the library's draw thread
    (safe_set_draw_cpus) and every other thread of the process is kept off them."""
    import os
    try:
        buf = C.create_string_buffer(32)
        check(lib.safe_device_pci_bus_id(int(device), buf, 32))
        bdf = buf.value.decode().lower()
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        mine = cpus
        if reserve_draw_cores:
            # whole physical cores for the library's draw thread: nothing else of this process then shares a core with it
            # (a polling launcher on the sibling hardware thread slows the sequential draw chain by a third)
            split = _split_off_cores(cpus, int(reserve_draw_cores), int(device))
            if split is not None:
                draw_cpus, mine = split
                arr = (C.c_int * len(draw_cpus))(*draw_cpus)
                check(lib.safe_set_draw_cpus(arr, len(draw_cpus)))
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), mine)
            except OSError:
                pass
        return node
    except Exception:
        return None


class DeviceBuffer:
    """A raw device allocation owned by the library allocator (f64/i64 element views).  Freed buffers
    of 1 MiB and more go back to a per-context pool keyed by size: hipMalloc / hipFree of the result
    matrices (GBs at 20 000 nodes) cost tens of milliseconds per call, more than the kernels.
    The pool is bounded by a share of the device's memory (POOL_FRACTION of ctx.hbm_bytes); when a
    returned buffer does not fit, the sizes that have gone unused the longest are released first, and an
    allocation that fails returns the whole pool to the driver and tries once more."""

    POOL_FRACTION = 1.0 / 6.0             # of the device memory (48 GB of 288 GB)

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        pooled = ctx._pool.get(self.nbytes)
        if pooled:
            self.ptr = pooled.pop()
            ctx._pool_bytes -= self.nbytes
            if pooled:
                ctx._pool.move_to_end(self.nbytes)         # most recently used size last
            else:
                del ctx._pool[self.nbytes]
            return
        p = C.c_void_p()
        rc = lib.safe_dev_alloc(ctx.handle, max(self.nbytes, 1), C.byref(p))
        if rc != 0 and ctx._pool_bytes:                    # out of memory with buffers idling in the pool: release them, retry
            ctx.trim()
            rc = lib.safe_dev_alloc(ctx.handle, max(self.nbytes, 1), C.byref(p))
        check(rc)
        self.ptr = p.value

    def free(self):
        if self.ptr:
            ctx = self.ctx
            limit = int(ctx.hbm_bytes * self.POOL_FRACTION)
            if (1 << 20) <= self.nbytes <= limit:
                while ctx._pool and ctx._pool_bytes + self.nbytes > limit:     # evict the least recently used size
                    size, ptrs = next(iter(ctx._pool.items()))
                    check(lib.safe_dev_free(ctx.handle, C.c_void_p(ptrs.pop())))
                    ctx._pool_bytes -= size
                    if not ptrs:
                        del ctx._pool[size]
                ctx._pool.setdefault(self.nbytes, []).append(self.ptr)
                ctx._pool.move_to_end(self.nbytes)
                ctx._pool_bytes += self.nbytes
            else:
                check(lib.safe_dev_free(ctx.handle, C.c_void_p(self.ptr)))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, host):
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        check(lib.safe_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), _ptr(host), host.nbytes))

    def download(self, shape, dtype=np.float64, out=None):
        """Copy to the host: into a fresh array, or into `out` (an array whose pages are resident -- it has been written
        before -- takes the plain copy at the link's rate; a fresh one the threaded copy that spreads its page faults)."""
        if out is not None:
            assert out.shape == tuple(shape) and out.dtype == dtype and out.flags.c_contiguous and out.nbytes <= self.nbytes
            check(lib.safe_memcpy_d2h_resident(self.ctx.handle, _ptr(out), C.c_void_p(self.ptr), out.nbytes))
            return out
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib.safe_memcpy_d2h(self.ctx.handle, _ptr(out), C.c_void_p(self.ptr), out.nbytes))
        return out

    def zero(self):
        check(lib.safe_dev_memset(self.ctx.handle, C.c_void_p(self.ptr), 0, self.nbytes))


class Context:
    _default = {}

    def __init__(self, device=0):
        h = C.c_void_p()
        check(lib.safe_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        ncu = C.c_int()
        hbm = C.c_int64()
        arch = C.create_string_buffer(64)
        check(lib.safe_ctx_info(h, C.byref(ncu), C.byref(hbm), arch, 64))
        self.num_cu = ncu.value
        self.hbm_bytes = hbm.value
        self.arch = arch.value.decode()
        self._pool = collections.OrderedDict()     # size -> [device pointers] of released DeviceBuffers, least recently used first
        self._pool_bytes = 0

    @classmethod
    def default(cls, device=0):
        """Process-wide context per device (created on first use)."""
        if device not in cls._default:
            cls._default[device] = cls(device)
        return cls._default[device]

    def set_stream(self, stream_ptr):
        check(lib.safe_ctx_set_stream(self.handle, C.c_void_p(stream_ptr) if stream_ptr else None))

    shared_stream = None        # (name, local_rank, local_world) once share_stream() has attached this context to a node's ring

    def share_stream(self, name, local_rank, local_world, capacity_bytes=64 << 20):
        """One permutation stream per node (safe_ctx_share_stream): local rank 0 draws, the other ranks of the node receive
        every chunk's row maps through a shared-memory ring.  Collective over the ranks of the node; afterwards
        Permutations(..., shared=True) on this context uses the ring."""
        if int(local_world) <= 1:
            return False
        check(lib.safe_ctx_share_stream(self.handle, str(name).encode(), int(local_rank), int(local_world), int(capacity_bytes)))
        self.shared_stream = (str(name), int(local_rank), int(local_world))
        return True

    def unshare_stream(self):
        check(lib.safe_ctx_unshare_stream(self.handle))
        self.shared_stream = None

    def sync(self):
        check(lib.safe_ctx_sync(self.handle))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def trim(self):
        """Return the pooled device buffers to the driver."""
        for ptrs in self._pool.values():
            for p in ptrs:
                check(lib.safe_dev_free(self.handle, C.c_void_p(p)))
        self._pool.clear()
        self._pool_bytes = 0

    def alloc_f64(self, *shape):
        return DeviceBuffer(self, int(np.prod(shape)) * 8)

    def timer_start(self):
        check(lib.safe_timer_start(self.handle))

    def timer_stop_ms(self):
        ms = C.c_double()
        check(lib.safe_timer_stop_ms(self.handle, C.byref(ms)))
        return ms.value

    def last_kernel(self):
        name = C.create_string_buffer(128)
        ms = C.c_double()
        cnt = C.c_int64()
        check(lib.safe_last_kernel_stats(self.handle, name, 128, C.byref(ms), C.byref(cnt)))
        return name.value.decode(), ms.value, cnt.value

    def last_kernel_busy_ms(self):
        """Time during which at least one launch of the dominant kernel of the last call was running (launches overlap)."""
        ms = C.c_double()
        check(lib.safe_last_kernel_busy_ms(self.handle, C.byref(ms)))
        return ms.value

    def edge_lengths(self, xy, edge_u, edge_v):
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        eu = np.ascontiguousarray(edge_u, dtype=np.int32)
        ev = np.ascontiguousarray(edge_v, dtype=np.int32)
        out = np.empty(eu.shape[0], dtype=np.float64)
        check(lib.safe_edge_lengths(self.handle, _ptr(xy), xy.shape[0], eu.shape[0], _ptr(eu), _ptr(ev), _ptr(out)))
        return out

    def euclidean_dense(self, xy_dev_ptr, n, nr, mask_dev_ptr=None, dist_dev_ptr=None):
        check(lib.safe_euclidean_dense_dev(self.handle, C.c_void_p(xy_dev_ptr), int(n), float(nr),
                                           C.c_void_p(mask_dev_ptr) if mask_dev_ptr else None,
                                           C.c_void_p(dist_dev_ptr) if dist_dev_ptr else None))


class Neighborhoods:
    """Device-resident membership (bit matrix + CSR + SELL-64)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.handle = handle
        n, nnz, mx = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.safe_nbr_info(handle, C.byref(n), C.byref(nnz), C.byref(mx)))
        self.n, self.nnz, self.max_row_count = n.value, nnz.value, mx.value

    @classmethod
    def euclidean(cls, ctx, xy, nr):
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        assert xy.ndim == 2 and xy.shape[1] == 2
        h = C.c_void_p()
        check(lib.safe_nbr_euclidean(ctx.handle, _ptr(xy), xy.shape[0], float(nr), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def shortpath(cls, ctx, n, edge_u, edge_v, edge_w, cutoff, keep_distances=False):
        eu = np.ascontiguousarray(edge_u, dtype=np.int32)
        ev = np.ascontiguousarray(edge_v, dtype=np.int32)
        ew = None if edge_w is None else np.ascontiguousarray(edge_w, dtype=np.float64)
        h = C.c_void_p()
        check(lib.safe_nbr_shortpath(ctx.handle, int(n), eu.shape[0], _ptr(eu), _ptr(ev), _ptr(ew), float(cutoff),
                                     1 if keep_distances else 0, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_dense(cls, ctx, a):
        a = np.asarray(a)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError('neighborhoods must be a square matrix, got shape %s' % (a.shape,))
        a = np.ascontiguousarray(a, dtype=np.int64)
        h = C.c_void_p()
        check(lib.safe_nbr_from_dense_i64(ctx.handle, _ptr(a), a.shape[0], C.byref(h)))
        return cls(ctx, h)

    def set_layout(self, xy):
        """Hint: the 2-D layout the membership came from (node order of the matrix-core kernel only)."""
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        assert xy.shape == (self.n, 2)
        check(lib.safe_nbr_set_layout(self.handle, _ptr(xy)))

    def to_dense(self):
        out = np.empty((self.n, self.n), dtype=np.int64)
        check(lib.safe_nbr_to_dense_i64(self.handle, _ptr(out)))
        return out

    def to_dense_dev(self, dev_ptr):
        check(lib.safe_nbr_to_dense_i64_dev(self.handle, C.c_void_p(dev_ptr)))

    def row_counts(self):
        out = np.empty(self.n, dtype=np.int64)
        check(lib.safe_nbr_row_counts(self.handle, _ptr(out)))
        return out

    def csr(self):
        rp = np.empty(self.n + 1, dtype=np.int32)
        col = np.empty(max(self.nnz, 1), dtype=np.int32)
        check(lib.safe_nbr_csr(self.handle, _ptr(rp), _ptr(col)))
        return rp, col[:self.nnz]

    def distances(self):
        out = np.empty((self.n, self.n), dtype=np.float64)
        check(lib.safe_nbr_distances(self.handle, _ptr(out)))
        return out

    def close(self):
        if self.handle:
            check(lib.safe_nbr_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Attributes:
    """Device-resident node x attribute matrix (f32/f64, C or Fortran order, NaN = missing)."""

    def __init__(self, ctx, handle, n, m, keepalive=None):
        self.ctx = ctx
        self.handle = handle
        self.n, self.m = n, m
        self._keepalive = keepalive

    @staticmethod
    def _layout(b, allow_u8=False):
        if b.ndim != 2:
            raise ValueError('node2attribute must be 2-D, got shape %s' % (b.shape,))
        if b.dtype == np.float32:
            dt = _lib.DTYPE_F32
        elif b.dtype == np.float64:
            dt = _lib.DTYPE_F64
        elif allow_u8 and b.dtype in (np.uint8, np.bool_):
            # additive: a 0/1 matrix as bytes travels as bytes (a quarter of the f32 upload) and is f32 on the device
            b = b.view(np.uint8)
            dt = _lib.DTYPE_U8
        else:
            b = b.astype(np.float64)
            dt = _lib.DTYPE_F64
        if not (b.flags['C_CONTIGUOUS'] or b.flags['F_CONTIGUOUS']):
            b = np.ascontiguousarray(b)
        n, m = b.shape
        if b.flags['C_CONTIGUOUS']:
            rs, cs = m, 1
        else:
            rs, cs = 1, n
        return b, dt, n, m, rs, cs

    @classmethod
    def from_host(cls, ctx, b):
        b, dt, n, m, rs, cs = cls._layout(np.asarray(b), allow_u8=True)
        h = C.c_void_p()
        check(lib.safe_attr_create_host(ctx.handle, _ptr(b), dt, n, m, rs, cs, C.byref(h)))
        return cls(ctx, h, n, m)

    @classmethod
    def from_device(cls, ctx, dev_ptr, dtype, n, m, order='C', keepalive=None):
        dt = _lib.DTYPE_F32 if np.dtype(dtype) == np.float32 else _lib.DTYPE_F64
        rs, cs = (m, 1) if order == 'C' else (1, n)
        h = C.c_void_p()
        check(lib.safe_attr_create_dev(ctx.handle, C.c_void_p(dev_ptr), dt, n, m, rs, cs, C.byref(h)))
        return cls(ctx, h, n, m, keepalive=keepalive)

    @classmethod
    def reindexed(cls, ctx, table, row_map, fill_value=np.nan, order='F', want_host=True):
        """read_attributes' alignment step (safe_io.py:386-390, 410): rows of the file's `table`
        gathered into node order on the device.  row_map[i] = table row of node i, -1 = absent
        (fill_value), -2 = masked duplicate (NaN).  Returns (handle, host copy or None); the host
        copy has the table's dtype and the requested memory order."""
        table, dt, nl, m, rs, cs = cls._layout(np.asarray(table))
        row_map = np.ascontiguousarray(row_map, dtype=np.int64)
        n = row_map.shape[0]
        host = np.empty((n, m), dtype=table.dtype, order=order) if want_host else None
        h = C.c_void_p()
        check(lib.safe_attr_reindex(ctx.handle, _ptr(table), dt, nl, m, rs, cs, _ptr(row_map), n, float(fill_value),
                                    0 if order == 'C' else 1, _ptr(host) if want_host else None, C.byref(h)))
        return cls(ctx, h, n, m), host

    def value_counts(self):
        """(#NaN, #zeros, #positives, #negatives) -- the census of safe_io.py:426-429."""
        v = [C.c_int64() for _ in range(4)]
        check(lib.safe_attr_value_counts(self.handle, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def nan_to_zero(self):
        """background='network' on the device copy (safe.py:449-451)."""
        check(lib.safe_attr_nan_to_zero(self.handle))

    def download(self, dtype, order='C'):
        out = np.empty((self.n, self.m), dtype=dtype, order=order)
        check(lib.safe_attr_download(self.handle, _ptr(out)))
        return out

    def stats(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.safe_attr_stats(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {'n_other': a.value, 'max_nan_col': b.value, 'n_rows_with_value': c.value, 'n_non_integer': d.value}

    def row_flags(self):
        out = np.empty(self.n, dtype=np.uint8)
        check(lib.safe_attr_row_flags(self.handle, _ptr(out)))
        return out

    def set_row_flags(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert flags.shape == (self.n,)
        check(lib.safe_attr_set_row_flags(self.handle, _ptr(flags)))

    def close(self):
        if self.handle:
            check(lib.safe_attr_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Permutations:
    """Device-resident composed row-permutation tables (legacy NumPy MT19937 stream)."""

    def __init__(self, ctx, n, movable, num_permutations, seed, shared=False, device_key=None):
        """seed: the reference's random_seed -- an integer reproduces np.random.seed(seed) + np.random.permutation exactly
        (safe_extras.py:46-58).  seed=None (the reference's default: OS entropy, nothing to reproduce) generates the tables
        ON THE DEVICE (safe_perms_create_device: i.i.d. uniform permutations, counter-based generator) keyed by `device_key`
        (64 bits; default: OS entropy) -- unless SAFE_HIP_DEVICE_STREAM=0 or more than 65535 rows move, then the NumPy-compatible
        stream runs from an entropy seed as before.
        shared=True: a COLLECTIVE call over the ranks of this node whose contexts share a stream (Context.share_stream):
        only the node's local rank 0 draws (its seed counts), the others receive the row maps.  Without a shared stream on
        the context it is the plain per-process stream.  (Device-generated tables need no sharing: every rank generates the
        same tables from the same key.)"""
        movable = np.ascontiguousarray(movable, dtype=np.uint8)
        assert movable.shape == (n,)
        if seed is not None:
            seed = int(seed)
            if seed < 0 or seed > 0xFFFFFFFF:
                raise ValueError('Seed must be between 0 and 2**32 - 1')
        self.ctx = ctx
        self.n = int(n)
        self.count = int(num_permutations)
        h = C.c_void_p()
        self.device_key = None
        if seed is None and device_stream_enabled() and int(movable.sum()) <= 65535:
            if shared and device_key is None:
                # a collective call: a key drawn here would differ from rank to rank and so would the tables
                raise ValueError('Permutations(seed=None, shared=True) needs a device_key the ranks agreed on '
                                 '(sharding.agree_on_seed / reduce_flags_and_stats)')
            self.device_key = int.from_bytes(os.urandom(8), 'little') if device_key is None else int(device_key) & 0xFFFFFFFFFFFFFFFF
            check(lib.safe_perms_create_device(ctx.handle, self.n, _ptr(movable), self.count, C.c_uint64(self.device_key), C.byref(h)))
        else:
            create = lib.safe_perms_create_shared if (shared and ctx.shared_stream) else lib.safe_perms_create
            check(create(ctx.handle, self.n, _ptr(movable), self.count, 0 if seed is None else 1,
                         0 if seed is None else seed, C.byref(h)))
        self.handle = h

    def timing(self):
        """Host-side timing of the stream (safe_perms_timing), ms, and this rank's role in it."""
        out = (C.c_double * 5)()
        check(lib.safe_perms_timing(self.handle, out))
        tw, ch, won = C.c_int(), C.c_int64(), C.c_int64()
        check(lib.safe_perms_twin_stats(self.handle, C.byref(tw), C.byref(ch), C.byref(won)))
        return {'draw_busy_ms': out[0], 'drawn_all_ms': out[1], 'tables_enqueued_ms': out[2], 'waited_for_producer_ms': out[3],
                'role': ('own', 'producer', 'consumer', 'device')[int(out[4])],
                'twin_chain': bool(tw.value), 'chunks': ch.value, 'chunks_won_by_twin': won.value}

    @classmethod
    def from_table(cls, ctx, perm_idx):
        """Tables supplied by the caller instead of the legacy stream (safe_perms_create_from_table): perm_idx is
        [P, n], row p the COMPOSED index vector -- permuted matrix p = B[perm_idx[p]] (safe_extras.py:58 applied
        cumulatively).  Every row must be a permutation of 0..n-1."""
        perm_idx = np.ascontiguousarray(perm_idx, dtype=np.int32)
        if perm_idx.ndim != 2:
            raise ValueError('perm_idx must be [num_permutations, n]')
        self = cls.__new__(cls)
        self.ctx = ctx
        self.count, self.n = int(perm_idx.shape[0]), int(perm_idx.shape[1])
        h = C.c_void_p()
        check(lib.safe_perms_create_from_table(ctx.handle, self.n, self.count, _ptr(perm_idx), C.byref(h)))
        self.handle = h
        return self

    def slice(self, p0, p1):
        """Permutations [p0, p1) as a handle of their own (safe_perms_slice): one rank's range of a permutation-axis split."""
        other = Permutations.__new__(Permutations)
        other.ctx, other.n, other.count = self.ctx, self.n, int(p1) - int(p0)
        h = C.c_void_p()
        check(lib.safe_perms_slice(self.handle, int(p0), int(p1), C.byref(h)))
        other.handle = h
        return other

    def read(self, p0=0, p1=None):
        p1 = self.count if p1 is None else p1
        out = np.empty((p1 - p0, self.n), dtype=np.int32)
        check(lib.safe_perms_read(self.handle, p0, p1, _ptr(out)))
        return out

    def close(self):
        if self.handle:
            check(lib.safe_perms_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator behind the C ABI (safe_comm_* / safe_allgather_cols): the final exchange of the
    attribute-sharded path for hosts without torch.distributed.  `Comm.unique_id()` on one rank, the 128 bytes to
    every rank by any means, then `Comm(ctx, world, rank, uid)` on each (one GPU per rank)."""

    ID_BYTES = 128

    @staticmethod
    def _framework_first():
        """When PyTorch-ROCm is installed its bundled RCCL must be the process's RCCL, loaded the way torch loads it:
        dlopen-ing that file ahead of `import torch` (or a second copy beside it) ends in a heap corruption at interpreter
        exit.  A host without torch uses the system's librccl (comm.cpp)."""
        try:
            import torch  # noqa: F401
        except ImportError:
            pass

    @staticmethod
    def unique_id():
        Comm._framework_first()
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(lib.safe_comm_unique_id(buf, Comm.ID_BYTES))
        return buf.raw

    def __init__(self, ctx, world_size, rank, unique_id):
        assert len(unique_id) == self.ID_BYTES
        self._framework_first()
        self.ctx, self.world_size, self.rank = ctx, int(world_size), int(rank)
        h = C.c_void_p()
        check(lib.safe_comm_create(ctx.handle, self.world_size, self.rank, unique_id, self.ID_BYTES, C.byref(h)))
        self.handle = h

    def allgather(self, local_ptr, bytes_per_rank, all_ptr):
        """Enqueue (context stream) the all-gather of every rank's `bytes_per_rank` bytes at device pointer
        `local_ptr` into `all_ptr` (world_size slabs in rank order)."""
        check(lib.safe_allgather_cols(self.handle, C.c_void_p(local_ptr), int(bytes_per_rank), C.c_void_p(all_ptr)))

    def close(self):
        if self.handle:
            check(lib.safe_comm_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_stream_enabled():
    """Unseeded runs generate their permutation tables on the device unless SAFE_HIP_DEVICE_STREAM=0."""
    return os.environ.get('SAFE_HIP_DEVICE_STREAM', '1') != '0'


def effective_cores():
    """CPUs this process may actually use: the scheduler affinity, capped by the container's CFS quota (cgroup v2 cpu.max /
    v1 cfs_quota_us) when there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                cores = min(cores, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return cores


def configure_host_for_ranks(local_world):
    """Host-side settings of one rank among `local_world` on this node, BEFORE its Context exists: when a rank has fewer than
    three cores to itself its host waits sleep instead of spinning (a spinning launcher + draw thread per rank on a 16-CPU
    quota gets the whole job throttled).  The only host thread of the permutation stream is the draw thread of a seeded
    call (one per node when the stream is shared); the swaps are replayed on the device.
    Environment override: SAFE_HIP_BLOCKING_SYNC.  Returns what was chosen."""
    local_world = max(1, int(local_world))
    share = effective_cores() / local_world
    blocking = os.environ.get('SAFE_HIP_BLOCKING_SYNC')
    blocking = (share < 3) if blocking is None else (blocking not in ('0', ''))
    set_blocking_sync(blocking)
    return {'cores_per_rank': share, 'blocking_sync': bool(blocking)}


def set_blocking_sync(on):
    """Host waits sleep on interrupt-backed events instead of spinning (safe_set_blocking_sync); before the first Context."""
    check(lib.safe_set_blocking_sync(1 if on else 0))


def device_alloc_count():
    """hipMalloc / hipHostMalloc calls the library has made in this process so far (safe_alloc_count)."""
    v = C.c_int64()
    check(lib.safe_alloc_count(C.byref(v)))
    return v.value


def last_mfma_slices(ctx):
    """i8 slices the last matrix-core permutation test ran with (2 / 4 / 6; 0 = it has not run)."""
    v = C.c_int()
    check(lib.safe_last_mfma_slices(ctx.handle, C.byref(v)))
    return v.value


def build_info():
    """Compiler version and build-time checks of the loaded library."""
    buf = C.create_string_buffer(512)
    check(lib.safe_build_info(buf, 512))
    return buf.value.decode()


def last_mfma_filter(ctx):
    """(slices the matrix cores multiplied, compares settled from the low digits) of the last matrix-core permutation test;
    a negative second value: the undecided list overflowed and the call was repeated with all six slices."""
    s, u = C.c_int(), C.c_int64()
    check(lib.safe_last_mfma_filter(ctx.handle, C.byref(s), C.byref(u)))
    return s.value, u.value


def rng_permutations_host(seed, values, count):
    """Host-only: `count` successive np.random.permutation(values) draws after np.random.seed(seed)."""
    values = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty((count, values.shape[0]), dtype=np.int64)
    check(lib.safe_rng_permutations_host(int(seed), _ptr(values), values.shape[0], int(count), _ptr(out)))
    return out


def nes_table(num_permutations):
    """-log10 of every possible empirical p-value k/P (k=0 -> 1/P), evaluated with NumPy so
    NES equals the reference's np.log10 bit for bit (safepy/safe.py:546-547)."""
    p = np.arange(num_permutations + 1, dtype=np.float64) / num_permutations
    with np.errstate(divide='ignore'):
        return np.ascontiguousarray(-np.log10(np.where(p == 0, 1 / num_permutations, p)))


# ---- enrichment entry points on device buffers -----------------------------------------

def score(ctx, nbr, attr, score_type, out_ptr, col0=0, col1=None):
    col1 = attr.m if col1 is None else col1
    check(lib.safe_score(ctx.handle, nbr.handle, attr.handle, _SCORE[score_type], col0, col1, C.c_void_p(out_ptr)))


def permtest_counts(ctx, nbr, attr, perms, score_type, ns_ptr, neg_ptr, pos_ptr, col0=0, col1=None):
    col1 = attr.m if col1 is None else col1
    check(lib.safe_permtest_counts(ctx.handle, nbr.handle, attr.handle, perms.handle, _SCORE[score_type], col0, col1,
                                   C.c_void_p(ns_ptr) if ns_ptr else None, C.c_void_p(neg_ptr), C.c_void_p(pos_ptr)))


def randomization(ctx, nbr, attr, perms, score_type, attribute_sign, enrichment_threshold, out_ptrs,
                  col0=0, col1=None, table=None):
    """out_ptrs = (ns, pvalues_neg, pvalues_pos, nes, nes_binary, num_enriched) device pointers."""
    col1 = attr.m if col1 is None else col1
    if table is None:
        table = nes_table(perms.count)
    ns, pn, pp, nes, nb, ne = out_ptrs
    check(lib.safe_randomization(ctx.handle, nbr.handle, attr.handle, perms.handle, _SCORE[score_type],
                                 _SIGN[attribute_sign], float(enrichment_threshold), _ptr(table), col0, col1,
                                 C.c_void_p(ns) if ns else None, C.c_void_p(pn), C.c_void_p(pp), C.c_void_p(nes),
                                 C.c_void_p(nb), C.c_void_p(ne)))


def outputs_from_counts(ctx, n, m, num_permutations, attribute_sign, enrichment_threshold, counts_neg_ptr, counts_pos_ptr, ns_ptr,
                        out_ptrs, table=None):
    """#<= / #>= counts of a whole call (f64 [n, m] on the device) -> out_ptrs = (pvalues_neg, pvalues_pos, nes, nes_binary,
    num_enriched) device pointers (safe_outputs_from_counts; safe.py:528-554, 468-472)."""
    if table is None:
        table = nes_table(num_permutations)
    pn, pp, nes, nb, ne = out_ptrs
    check(lib.safe_outputs_from_counts(ctx.handle, int(n), int(m), int(num_permutations), _SIGN[attribute_sign],
                                       float(enrichment_threshold), _ptr(table), C.c_void_p(counts_neg_ptr), C.c_void_p(counts_pos_ptr),
                                       C.c_void_p(ns_ptr) if ns_ptr else None, C.c_void_p(pn), C.c_void_p(pp), C.c_void_p(nes),
                                       C.c_void_p(nb), C.c_void_p(ne)))


def hypergeom(ctx, nbr, attr, enrichment_threshold, out_ptrs, col0=0, col1=None):
    """out_ptrs = (pvalues_pos, nes, nes_binary, num_enriched) device pointers."""
    col1 = attr.m if col1 is None else col1
    pp, nes, nb, ne = out_ptrs
    check(lib.safe_hypergeom(ctx.handle, nbr.handle, attr.handle, float(enrichment_threshold), col0, col1,
                             C.c_void_p(pp), C.c_void_p(nes), C.c_void_p(nb), C.c_void_p(ne)))


def packed_counts_info(ctx):
    """(n_pad, m, layout) of the integer counters the last randomization call left on the device;
    layout -1 = none (f64 kernels)."""
    n_pad, m, layout = C.c_int64(), C.c_int64(), C.c_int()
    check(lib.safe_export_packed_counts(ctx.handle, None, 0, C.byref(n_pad), C.byref(m), C.byref(layout)))
    return n_pad.value, m.value, layout.value


def export_packed_counts(ctx, dst_ptr, capacity):
    n_pad, m, layout = C.c_int64(), C.c_int64(), C.c_int()
    check(lib.safe_export_packed_counts(ctx.handle, C.c_void_p(dst_ptr), int(capacity), C.byref(n_pad), C.byref(m),
                                        C.byref(layout)))
    return n_pad.value, m.value, layout.value


_ENQUEUED_FN = C.CFUNCTYPE(None, C.c_void_p)


def set_exchange_chunks(ctx, chunks, cols_per_chunk=0, on_enqueued=None):
    """Arms (chunks >= 2) or switches off (0) the column-chunked tail of the following randomization calls on `ctx`
    (safe_set_exchange_chunks).  `on_enqueued()` is called on the calling thread, inside the randomization call, once all
    of its launches are enqueued.  An exception it raises is kept and re-raised by take_exchange_error()."""
    if not chunks:
        check(lib.safe_set_exchange_chunks(ctx.handle, 0, 0, None, None))
        ctx._xc_cb = None
        return
    ctx._xc_error = None

    def trampoline(_user):
        try:
            if on_enqueued is not None:
                on_enqueued()
        except BaseException as err:           # (an exception cannot cross the C frames of the call)
            ctx._xc_error = err
    cb = _ENQUEUED_FN(trampoline)
    ctx._xc_cb = cb                             # the library keeps the raw pointer: the object must outlive the calls
    check(lib.safe_set_exchange_chunks(ctx.handle, int(chunks), int(cols_per_chunk), C.cast(cb, C.c_void_p), None))


def take_exchange_error(ctx):
    err, ctx._xc_error = getattr(ctx, '_xc_error', None), None
    if err is not None:
        raise err


def packed_chunk_info(ctx):
    """(chunks, [column bounds], tail permutations) of the last randomization call; chunks = 0: it ran no column-chunked launches."""
    k, tail = C.c_int(), C.c_int64()
    bounds = (C.c_int64 * 9)()
    check(lib.safe_packed_chunk_info(ctx.handle, C.byref(k), bounds, C.byref(tail)))
    return k.value, [bounds[i] for i in range(k.value + 1)] if k.value else [], tail.value


PACKED_NARROW = 16          # safe_hip.h SAFE_PACKED_NARROW: slabs of 20-bit counter pairs (num_permutations <= 1023)


def packed_slab_words(cols, n_pad, narrow):
    """u32 words of a slab of `cols` columns of packed counters: n_pad per column, or 5 n_pad / 8 in the narrow form."""
    return int(cols) * (int(n_pad) // 8 * 5 if narrow else int(n_pad))


def export_packed_chunk(ctx, chunk, dst_ptr, capacity, stream=None, narrow=False):
    """Chunk `chunk` of the last randomization call's counters to dst_ptr (`capacity` u32 words, the rest zeroed) on `stream`;
    narrow: two outputs in five bytes (safe_export_packed_chunk_narrow, num_permutations <= 1023)."""
    fn = lib.safe_export_packed_chunk_narrow if narrow else lib.safe_export_packed_chunk
    check(fn(ctx.handle, int(chunk), C.c_void_p(dst_ptr), int(capacity), C.c_void_p(stream) if stream else None))


def outputs_from_packed_slabs(ctx, nbr, slabs_ptr, layout, n_pad, slab_stride, slab_cols, out_col0, m_total, num_permutations,
                              attribute_sign, enrichment_threshold, out_ptrs, table=None, stream=None):
    """out_ptrs = (pvalues_neg, pvalues_pos, nes, nes_binary) device pointers of f64 [n, m_total] matrices (None = not wanted);
    slab r = slab_cols[r] columns of u32 counters at slabs_ptr + 4 * r * slab_stride, written to columns out_col0[r] ...
    (safe_outputs_from_packed_slabs); enqueued on `stream` (a raw hipStream_t; None: the context's), not waited for."""
    if table is None:
        table = nes_table(num_permutations)
    k = len(slab_cols)
    cols = (C.c_int64 * k)(*[int(v) for v in slab_cols])
    col0 = (C.c_int64 * k)(*[int(v) for v in out_col0])
    pn, pp, nes, nb = (C.c_void_p(p) if p else None for p in out_ptrs)
    check(lib.safe_outputs_from_packed_slabs(ctx.handle, nbr.handle, C.c_void_p(slabs_ptr), int(layout), int(n_pad), k, int(slab_stride),
                                             cols, col0, int(m_total), int(num_permutations), _SIGN[attribute_sign],
                                             float(enrichment_threshold), _ptr(table), pn, pp, nes, nb,
                                             C.c_void_p(stream) if stream else None))


def randomization_plan(ctx, nbr, attr, num_permutations, score_type):
    """The packed-counter layout safe_randomization would leave for this block (0 = bit-sliced kernel), -1 = none predicted."""
    layout = C.c_int()
    check(lib.safe_randomization_plan(ctx.handle, nbr.handle, attr.handle, int(num_permutations), _SCORE[score_type], C.byref(layout)))
    return layout.value


def nes_from_packed_counts(ctx, nbr, counts_ptr, layout, n_pad, m, num_permutations, attribute_sign, nes_ptr, table=None):
    if table is None:
        table = nes_table(num_permutations)
    check(lib.safe_nes_from_packed_counts(ctx.handle, nbr.handle, C.c_void_p(counts_ptr), int(layout), int(n_pad), int(m),
                                          int(num_permutations), _SIGN[attribute_sign], _ptr(table), C.c_void_p(nes_ptr)))


def outputs_from_packed_counts(ctx, nbr, counts_ptr, layout, n_pad, m, num_permutations, attribute_sign, enrichment_threshold,
                               out_ptrs, table=None):
    """out_ptrs = (pvalues_neg, pvalues_pos, nes, nes_binary) device pointers of f64 [n, m] matrices, None = not wanted
    (safe_outputs_from_packed_counts)."""
    if table is None:
        table = nes_table(num_permutations)
    pn, pp, nes, nb = (C.c_void_p(p) if p else None for p in out_ptrs)
    check(lib.safe_outputs_from_packed_counts(ctx.handle, nbr.handle, C.c_void_p(counts_ptr), int(layout), int(n_pad), int(m),
                                              int(num_permutations), _SIGN[attribute_sign], float(enrichment_threshold), _ptr(table),
                                              pn, pp, nes, nb))


def block_count(nbr):
    """Stored 256 x 32 blocks of the block-sparse membership the matrix-core kernel multiplies
    (built on first use by that kernel; 0 before)."""
    c = C.c_int64()
    check(lib.safe_nbr_block_count(nbr.handle, C.byref(c)))
    return c.value


def piece_count(nbr):
    """32 x 32 pieces of the stored membership blocks that hold a member: the ones the matrix-core kernel multiplies (safe_nbr_piece_count)."""
    v = C.c_int64()
    check(lib.safe_nbr_piece_count(nbr.handle, C.byref(v)))
    return v.value


def fdr_adjust(ctx, n, m, num_permutations, attribute_sign, enrichment_threshold, out_ptrs):
    """multiple_testing=True: out_ptrs = (pvalues_neg or None, pvalues_pos, nes, nes_binary, num_enriched);
    num_permutations = 0 selects the hypergeometric form."""
    pn, pp, nes, nb, ne = out_ptrs
    check(lib.safe_fdr_adjust(ctx.handle, int(n), int(m), int(num_permutations), _SIGN[attribute_sign],
                              float(enrichment_threshold), C.c_void_p(pn) if pn else None, C.c_void_p(pp), C.c_void_p(nes),
                              C.c_void_p(nb), C.c_void_p(ne)))


def enriched_components(ctx, n, edge_u, edge_v, member):
    """member: [n, n_cols] (> 0 = enriched).  Returns int32 [n_cols, n] component labels (smallest
    node id of the component; -1 = not enriched) -- safe.py:640-655 for all attributes at once."""
    member = np.ascontiguousarray(member, dtype=np.float64)
    assert member.ndim == 2 and member.shape[0] == n
    eu = np.ascontiguousarray(edge_u, dtype=np.int32)
    ev = np.ascontiguousarray(edge_v, dtype=np.int32)
    out = np.empty((member.shape[1], n), dtype=np.int32)
    check(lib.safe_enriched_components(ctx.handle, int(n), eu.shape[0], _ptr(eu), _ptr(ev), _ptr(member), member.shape[1], _ptr(out)))
    return out


def jaccard_condensed(ctx, x):
    """scipy pdist(x, 'jaccard') (condensed) for the rows of x [m_top, n]."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    m_top, n = x.shape
    out = np.empty(m_top * (m_top - 1) // 2, dtype=np.float64)
    check(lib.safe_jaccard_condensed(ctx.handle, m_top, n, _ptr(x), _ptr(out)))
    return out
