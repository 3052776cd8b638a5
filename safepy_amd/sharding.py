"""Attribute sharding of compute_pvalues across the GPUs of a node.

The reference's only multi-process pattern (safepy/safe.py:1321-1361) splits the attribute
columns with `np.array_split` over `cpu_count()` worker processes, runs the whole pipeline in
each and concatenates the NES blocks (`np.concatenate(axis=1)`, safe.py:1355).  The same
decomposition here: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm),
rank r owns the column block `column_shards(m, world)[r]`; the membership and the seeded
permutation stream are replicated (same seed => identical tables, no broadcast).  The path has
exactly two exchange steps:

  * whole-matrix facts that must NOT be computed per shard -- which rows carry a value
    (`indx_vals`, safe_extras.py:51; hypergeometric population, safe.py:574-578) and the
    dispatch / warning statistics (safe.py:453-463): an all-reduce (MAX / SUM) of a few bytes;
  * the final all-gather of every rank's [N, M_r] result block (the `np.concatenate`).

The collective helpers work on CPU tensors with the gloo backend as well, which is how the
N > 1 plumbing is tested without GPUs (tests/test_sharding_gloo.py).
"""
import numpy as np


def column_shards(m, world):
    """[(c0, c1)] per rank, np.array_split semantics (safe.py:1339)."""
    base, extra = divmod(int(m), int(world))
    out, c0 = [], 0
    for r in range(world):
        c1 = c0 + base + (1 if r < extra else 0)
        out.append((c0, c1))
        c0 = c1
    return out


def _dist():
    import torch.distributed as dist
    return dist


def _device_for(group=None):
    import torch
    dist = _dist()
    backend = dist.get_backend(group)
    return torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')


def reduce_row_flags(local_flags, group=None):
    """Rows holding at least one value in ANY rank's columns: element-wise MAX all-reduce of the
    per-shard uint8 flags.  Returns a NumPy uint8 array (identical on every rank)."""
    import torch
    dist = _dist()
    t = torch.from_numpy(np.array(local_flags, dtype=np.uint8, copy=True)).to(_device_for(group))   # never aliases the input
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy()


def reduce_stats(local_stats, group=None):
    """Whole-matrix statistics from per-shard ones: n_other / n_non_integer add up, the worst
    NaN column is a maximum (safe.py:453-463).  `n_rows_with_value` must come from
    reduce_row_flags, not from here."""
    import torch
    dist = _dist()
    dev = _device_for(group)
    sums = torch.tensor([local_stats['n_other'], local_stats['n_non_integer']], dtype=torch.int64, device=dev)
    mx = torch.tensor([local_stats['max_nan_col']], dtype=torch.int64, device=dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return {'n_other': int(sums[0]), 'n_non_integer': int(sums[1]), 'max_nan_col': int(mx[0])}


def reduce_flags_and_stats(local_flags, local_stats, group=None, random_seed=0, agree=1):
    """reduce_row_flags + reduce_stats in ONE collective: every rank contributes
    [flags (n), n_other, n_non_integer, max_nan_col, seed, agree] as int64, the all-gathered [world, n+5]
    table is reduced locally (MAX over flags and the worst NaN column, SUM over the counts, MIN over `agree`).
    Returns (flags uint8 [n], stats dict), identical on every rank.  stats['random_seed'] is the
    seed all ranks must use: `random_seed` itself when it is given, else ONE value drawn from OS
    entropy by rank 0 -- the reference permutes whole rows, one stream shared by all attributes
    (safe_extras.py:46, 58), so ranks that each seeded themselves would not be a split of one run.
    stats['agree'] = the smallest `agree` of any rank: what the ranks settle BEFORE the kernels run without a collective of
    its own (randomization_step: 1 = this rank's block will leave bit-sliced counters and can exchange them chunk by chunk)."""
    import torch
    dist = _dist()
    dev = _device_for(group)
    world = dist.get_world_size(group)
    n = len(local_flags)
    # NumPy on the host side on purpose: torch CPU kernels may open an OpenMP region whose
    # workers then spin (hundreds of CPU-milliseconds per call) -- under a container CPU quota
    # that gets the whole process throttled for the rest of the scheduler period
    mine = np.empty(n + 5, dtype=np.int64)
    mine[:n] = np.asarray(local_flags, dtype=np.int64)
    seed = _entropy63() if random_seed is None else int(random_seed)
    mine[n:] = (int(local_stats['n_other']), int(local_stats['n_non_integer']), int(local_stats['max_nan_col']), seed, int(agree))
    mine = torch.from_numpy(mine).to(dev)
    table = torch.empty(world * (n + 5), dtype=torch.int64, device=dev)     # flat: gloo wants 1-D buffers
    dist.all_gather_into_tensor(table, mine, group=group)
    table = table.cpu().numpy().reshape(world, n + 5)
    flags = table[:, :n].max(axis=0).astype(np.uint8)
    stats = {'n_other': int(table[:, n].sum()), 'n_non_integer': int(table[:, n + 1].sum()),
             'max_nan_col': int(table[:, n + 2].max()), 'random_seed': int(table[0, n + 3]), 'agree': int(table[:, n + 4].min())}
    return flags, stats


def _entropy63():
    """63 bits of OS entropy (fits the int64 tensors the ranks exchange): the key of an unseeded run's device stream.  Where such a
    run falls back to the NumPy-compatible stream (more than 65535 movable rows, SAFE_HIP_DEVICE_STREAM=0) its low 32 bits seed it."""
    import os
    return int.from_bytes(os.urandom(8), 'little') >> 1


def agree_on_seed(random_seed, group=None):
    """The seed every rank uses: `random_seed` when given, else rank 0's draw from OS entropy."""
    import torch
    dist = _dist()
    seed = _entropy63() if random_seed is None else int(random_seed)
    t = torch.tensor([seed], dtype=torch.int64, device=_device_for(group))
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def ensure_shared_stream(ctx, group=None):
    """One permutation stream per NODE instead of one per rank (every rank drew the whole MT19937 stream --
    safe_extras.py:46-58 is sequential, a rank cannot draw "its part" -- which made 8 ranks keep 40 host threads busy and
    capped strong scaling at the serial draw rate).  On first use (collective over the default group) the ranks agree on a
    job-unique name (one 8-byte broadcast from rank 0) and attach their context to the node's shared-memory ring
    (Context.share_stream): local rank 0 draws and replays, the others block until a chunk's row maps are published.
    Local rank / local world come from the launcher (LOCAL_RANK / LOCAL_WORLD_SIZE, set by torch.distributed.run); without
    them SAFE_HIP_SHARED_STREAM=1 declares all ranks to be on one node.  SAFE_HIP_SHARED_STREAM=0 keeps one stream per
    rank.  Returns True when this context shares a stream."""
    import os
    if ctx is None:
        return False
    if ctx.shared_stream is not None:
        return True
    if getattr(ctx, '_shared_stream_checked', False):
        return False
    ctx._shared_stream_checked = True
    dist = _dist()
    mode = os.environ.get('SAFE_HIP_SHARED_STREAM', '')
    if mode == '0' or group is not None or not dist.is_initialized() or dist.get_world_size() <= 1:
        return False
    if 'LOCAL_WORLD_SIZE' in os.environ and 'LOCAL_RANK' in os.environ:
        local_rank, local_world = int(os.environ['LOCAL_RANK']), int(os.environ['LOCAL_WORLD_SIZE'])
    elif mode == '1':
        local_rank, local_world = dist.get_rank(), dist.get_world_size()
    else:
        return False
    import torch
    token = torch.tensor([int.from_bytes(os.urandom(7), 'little')], dtype=torch.int64, device=_device_for(group))
    dist.broadcast(token, src=0)
    try:
        ok = ctx.share_stream('%014x' % int(token.item()), local_rank, local_world)
    except Exception as err:                           # no /dev/shm, no room, a local rank 0 that never showed up ...
        import logging
        logging.warning('safepy_amd: rank %d could not join the node\'s shared permutation stream (%s); every rank of the job '
                        'draws its own stream from now on', dist.get_rank(), err)
        ok = False
    # all or nothing: a node whose ring could not be set up falls back to one stream per rank -- on EVERY rank, or the
    # collective calls that follow would disagree about who draws
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=_device_for(group))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if ctx.shared_stream is not None:
            ctx.unshare_stream()
        return False
    return True


COUNTER_OUTPUTS = ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary')     # what the integer counters determine (safe.py:532-554, 468-472)


def gather_outputs(ctx, nbr, bufs, names, m_total, num_permutations, attribute_sign, enrichment_threshold=0.05, group=None,
                   table=None, report=None):
    """The final exchange (np.concatenate(axis=1) of every rank's block, safe.py:1355; north star: "final RCCL all-gather
    of the p-value matrix") for any of COUNTER_OUTPUTS, on every rank, from ONE all-gather of integers.

    When the last randomization call on every rank left packed integer counters on the device (bit-sliced / matrix-core
    'sum' kernels), the ranks all-gather those -- u32 per (node, attribute) instead of one f64 per requested matrix,
    attribute-major so a rank's block is one contiguous slab and no transpose copy is needed -- and each rank derives the
    requested full [N, M] matrices from the counters with the arithmetic of safe.py:532-554 and 468-472
    (safe_outputs_from_packed_counts).  Otherwise (f64 kernels, z-scores, no device) the f64 blocks travel
    (gather_columns).  `bufs`: this rank's [N, M_r] device tensors by name (their values are only used on the fallback
    path).  Over RCCL the slabs move device to device; over gloo (tests: several ranks on one GPU) they are staged through
    the host.  Returns {name: full tensor}; `report` (a dict) receives the form and the bytes this rank received."""
    import torch
    from . import backend as be
    dist = _dist()
    names = tuple(names)
    assert names and all(k in COUNTER_OUTPUTS for k in names), names
    world = dist.get_world_size(group)
    local = bufs[names[0]]
    on_device = ctx is not None and local.is_cuda
    n_pad, m_loc, layout = be.packed_counts_info(ctx) if on_device else (0, 0, -1)
    shards = column_shards(m_total, world)
    widest = max(c1 - c0 for c0, c1 in shards)
    dev = local.device
    xdev = _device_for(group)                          # where the collective's buffers live
    # every rank must take the same branch: agree on (layout, n_pad) with one tiny MIN reduce
    key = torch.tensor([layout, -layout, n_pad, -n_pad], dtype=torch.int64, device=xdev)
    dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
    key = key.tolist()
    agreed = key[0] >= 0 and key[0] == -key[1] and key[2] == -key[3]
    if not agreed:
        if report is not None:
            report.update(form='f64 blocks', bytes_received=int(8 * local.shape[0] * widest * (world - 1) * len(names)))
        return {k: gather_columns(bufs[k], m_total, group) for k in names}
    mine = torch.zeros(widest * n_pad, dtype=torch.int32, device=dev)       # zero counters = padding columns
    torch.cuda.current_stream().synchronize()          # the fill ran on torch's stream, the export runs on the context's
    be.export_packed_counts(ctx, mine.data_ptr(), m_loc * n_pad)
    ctx.sync()                                         # the export ran on the context's stream
    if xdev.type == 'cuda':
        everyone = torch.empty(world * widest * n_pad, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(everyone, mine, group=group)
    else:
        staged = torch.empty(world * widest * n_pad, dtype=torch.int32)
        dist.all_gather_into_tensor(staged, mine.cpu(), group=group)
        everyone = staged.to(dev)
    if not all(c1 - c0 == widest for c0, c1 in shards):           # ragged split: drop the padding columns of the narrower ranks
        slabs = everyone.view(world, widest, n_pad)
        everyone = torch.cat([slabs[r, :shards[r][1] - shards[r][0]] for r in range(world)], dim=0).contiguous()
    full = {k: torch.empty((local.shape[0], m_total), dtype=torch.float64, device=dev) for k in names}
    torch.cuda.current_stream().synchronize()          # the collective is ordered on torch's stream
    be.outputs_from_packed_counts(ctx, nbr, everyone.data_ptr(), layout, n_pad, m_total, num_permutations, attribute_sign,
                                  enrichment_threshold, [full[k].data_ptr() if k in full else None for k in COUNTER_OUTPUTS],
                                  table=table)
    ctx.sync()
    if report is not None:
        report.update(form='packed u32 counters (4 B per node x attribute), %s rebuilt on every rank' % ' / '.join(names),
                      bytes_received=int(4 * n_pad * widest * (world - 1)))
    return full


def exchange_chunk_grid(m_total, world):
    """(chunks, columns per chunk) of the chunked exchange for a run of m_total attributes over `world` ranks.
    SAFE_HIP_XCHG_CHUNKS: 1..8 chunks (default 1: the whole block in one slab), fewer when the widest block has fewer than two
    64-column word groups per chunk."""
    import os
    want = max(1, min(8, int(os.environ.get('SAFE_HIP_XCHG_CHUNKS', '1'))))
    widest = max(c1 - c0 for c0, c1 in column_shards(m_total, world))
    groups = max(1, -(-widest // 64))
    chunks = max(1, min(want, groups // 2))
    return chunks, 64 * (-(-groups // chunks))


class ChunkedExchange:
    """The final exchange of randomization_step when every rank's block runs the bit-sliced kernel -- settled BEFORE the kernels in
    the head collective (reduce_flags_and_stats: `agree`), so no collective and no host synchronisation is spent on agreeing
    afterwards (gather_outputs: one MIN all-reduce and two stream syncs per step).

    The counters travel in `chunks` column slabs (default 1).  Each slab is copied out on a side stream and all-gathered
    (RCCL: asynchronous, its own stream; gloo: staged through the host); the f64 matrices of chunk k are derived while chunk
    k + 1 travels (safe_outputs_from_packed_slabs writes column blocks of the full matrices).  One word travels with every slab:
    1 = bit-sliced counters; a rank whose call took another form after all still takes part in the same collectives and flags
    its slabs 0, and every rank then falls back to the f64 exchange (gather_columns) together.  One host wait, at the end.

    With SAFE_HIP_XCHG_TAIL set and >= 2 chunks the library also runs the last permutations one column chunk at a time and calls
    `launched()` from inside the randomization call; the chunks then travel while the later ones compute.  Measured at
    configs[1]: costs the step more than it hides -- off by default."""
    HEADER = 64                     # u32 words behind a slab (one used; keeps slabs 256-byte aligned)

    def __init__(self, ctx, nbr, bufs, names, m_total, num_permutations, attribute_sign, enrichment_threshold, group, table,
                 chunks, cols):
        import torch
        self.ctx, self.nbr, self.bufs, self.names, self.m_total = ctx, nbr, bufs, tuple(names), int(m_total)
        self.P, self.sign, self.thr, self.group, self.table = int(num_permutations), attribute_sign, enrichment_threshold, group, table
        self.chunks, self.cols = int(chunks), int(cols)
        dist = _dist()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.shards = column_shards(self.m_total, self.world)
        self.dev = bufs[self.names[0]].device
        self.xdev = _device_for(group)
        self.n = bufs[self.names[0]].shape[0]
        self.n_pad = 64 * (-(-self.n // 64))                 # SELL-64 positions: the bit-sliced kernel's counter rows
        # P <= 1023: the counter pair is 10 + 10 bits, two outputs travel in five bytes (safe_export_packed_chunk_narrow) --
        # 0.625 of the u32 slab; more permutations keep u32 pairs
        from . import backend as be
        self.narrow = self.P <= 1023
        self.cap = be.packed_slab_words(self.cols, self.n_pad, self.narrow) + self.HEADER
        # ONE side stream per context: torch hands out pooled streams round-robin, and its caching allocator keeps a pool per stream
        # -- a fresh stream per step meant fresh hipMallocs of every slab per step (the step doubled)
        side = getattr(ctx, '_xc_side', None)
        if side is None or side.device != self.dev:
            side = ctx._xc_side = torch.cuda.Stream(device=self.dev)
        self.side = side
        self.mine = [None] * self.chunks
        self.every = [None] * self.chunks
        self.work = [None] * self.chunks
        self.issued = False
        self.t_launched = None

    def arm(self):
        from . import backend as be
        be.set_exchange_chunks(self.ctx, self.chunks, self.cols, self.launched)

    def disarm(self):
        from . import backend as be
        be.set_exchange_chunks(self.ctx, 0)

    def _issue(self, valid):
        """Copy-out + all-gather of every chunk, in order; `valid`: this rank's counters are bit-sliced ones."""
        import time
        import torch
        from . import backend as be
        dist = _dist()
        self.t_launched = time.perf_counter()
        with torch.cuda.stream(self.side):
            for k in range(self.chunks):
                if valid:
                    mine = torch.empty(self.cap, dtype=torch.int32, device=self.dev)
                    be.export_packed_chunk(self.ctx, k, mine.data_ptr(), self.cap - self.HEADER, self.side.cuda_stream, narrow=self.narrow)
                    mine[self.cap - self.HEADER:].fill_(1)
                else:                                      # zero counters decode to in-range table entries on every rank
                    mine = torch.zeros(self.cap, dtype=torch.int32, device=self.dev)
                if self.xdev.type == 'cuda':
                    every = torch.empty(self.world * self.cap, dtype=torch.int32, device=self.dev)
                    self.work[k] = dist.all_gather_into_tensor(every, mine, group=self.group, async_op=True)
                else:
                    every = torch.empty(self.world * self.cap, dtype=torch.int32)
                    mine = mine.cpu()                      # (waits for this chunk's launch: tests and one-GPU rehearsals only)
                    self.work[k] = dist.all_gather_into_tensor(every, mine, group=self.group, async_op=True)
                self.mine[k], self.every[k] = mine, every
        self.issued = True

    def launched(self):
        """Called inside the randomization call, all launches enqueued."""
        from . import backend as be
        made, _bounds, _tail = be.packed_chunk_info(self.ctx)
        if made == self.chunks:
            self._issue(True)

    def finish(self, report=None):
        """After the randomization call: {name: full [N, m_total] device tensor} on every rank."""
        import torch
        from . import backend as be
        be.take_exchange_error(self.ctx)
        if not self.issued:                    # the call ran no chunked tail here: the same collectives, from the finished counters
            self._issue(be.packed_counts_info(self.ctx)[2] == 0)
        full = {k: torch.empty((self.n, self.m_total), dtype=torch.float64, device=self.dev) for k in self.names}
        ptrs = [full[k].data_ptr() if k in full else None for k in COUNTER_OUTPUTS]
        headers = []
        with torch.cuda.stream(self.side):
            for k in range(self.chunks):
                self.work[k].wait()            # RCCL: the side stream waits; gloo: the host does
                every = self.every[k] if self.every[k].is_cuda else self.every[k].to(self.dev, non_blocking=True)
                self.every[k] = every
                slabs = every.view(self.world, self.cap)
                headers.append(slabs[:, self.cap - self.HEADER])
                cols = [max(0, min(self.cols if k + 1 < self.chunks else c1 - c0, c1 - c0 - k * self.cols)) for c0, c1 in self.shards]
                col0 = [min(c0 + k * self.cols, c1) for c0, c1 in self.shards]      # (an empty trailing chunk starts at the block's end)
                be.outputs_from_packed_slabs(self.ctx, self.nbr, every.data_ptr(), be.PACKED_NARROW if self.narrow else 0, self.n_pad,
                                             self.cap, cols, col0, self.m_total,
                                             self.P, self.sign, self.thr, ptrs, table=self.table, stream=self.side.cuda_stream)
            ok = bool((torch.stack(headers) == 1).all().item())          # (the one host wait: everything above has finished)
        if report is not None:
            report.update(form='packed %s counters in %d column chunk%s (agreed before the kernels), %s rebuilt on every rank'
                               % ('10 + 10 bit (2.5 B per node x attribute)' if self.narrow else 'u32', self.chunks,
                                  '' if self.chunks == 1 else 's', ' / '.join(self.names)),
                          bytes_received=int(4 * self.cap * self.chunks * (self.world - 1)), chunks=self.chunks)
        if not ok:
            if report is not None:
                report.update(form='f64 blocks (a rank left no bit-sliced counters)')
            return {k: gather_columns(self.bufs[k], self.m_total, self.group) for k in self.names}
        return full


def gather_nes(ctx, nbr, local_nes, m_total, num_permutations, attribute_sign, group=None, table=None):
    """The NES matrix alone (gather_outputs with names=('nes',))."""
    return gather_outputs(ctx, nbr, {'nes': local_nes}, ('nes',), m_total, num_permutations, attribute_sign, 0.05, group, table)['nes']


def gather_columns(local, m_total, group=None):
    """All-gather of the per-rank [N, M_r] blocks into the full [N, M] matrix on every rank
    (the reference's np.concatenate(axis=1), safe.py:1355).  `local` is a 2-D torch tensor;
    blocks may differ by one column (array_split), so they travel padded to the widest block.
    A device tensor under a host backend (gloo) is staged through the host and comes back on
    its device."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    shards = column_shards(m_total, world)
    widest = max(c1 - c0 for c0, c1 in shards)
    n = local.shape[0]
    home = local.device
    xdev = _device_for(group)
    if local.shape[1] != widest:
        padded = torch.zeros((n, widest), dtype=local.dtype, device=home)
        padded[:, :local.shape[1]] = local
    else:
        padded = local.contiguous()
    if padded.device.type != xdev.type:
        padded = padded.to(xdev)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    full = torch.cat([parts[r][:, :shards[r][1] - shards[r][0]] for r in range(world)], dim=1)
    return full if full.device == home else full.to(home)


def _fdr_whole_matrix(ctx, bufs, m_total, num_permutations, attribute_sign, enrichment_threshold, group, ready=None):
    """multiple_testing=True under attribute sharding, with whole-matrix semantics: Benjamini-Hochberg runs
    along a ROW across all attributes (safe.py:536-542, 599-605), so the p-value blocks are all-gathered
    first and every rank adjusts the full [N, M] matrices (safe_fdr_adjust also rebuilds NES, nes_binary and
    the enriched counts from the adjusted values, safe.py:546-554 / 608 and 468-472).  `bufs`: this rank's
    device blocks by name; returns the full device tensors by name (+ 'num_neighborhoods_enriched')."""
    import torch
    from . import backend as be
    ready = ready or {}              # full unadjusted p-value matrices the integer exchange already produced
    full = {k: (ready[k] if k in ready else gather_columns(bufs[k], m_total, group)).contiguous()
            for k in ('pvalues_neg', 'pvalues_pos') if k in bufs}
    n = full['pvalues_pos'].shape[0]
    dev = full['pvalues_pos'].device
    full['nes'] = torch.empty((n, m_total), dtype=torch.float64, device=dev)
    full['nes_binary'] = torch.empty((n, m_total), dtype=torch.float64, device=dev)
    full['num_neighborhoods_enriched'] = torch.empty((m_total,), dtype=torch.float64, device=dev)
    torch.cuda.current_stream().synchronize()          # gathered on torch's stream, adjusted on the context's
    be.fdr_adjust(ctx, n, m_total, int(num_permutations), attribute_sign, enrichment_threshold,
                  [full['pvalues_neg'].data_ptr() if 'pvalues_neg' in full else None, full['pvalues_pos'].data_ptr(),
                   full['nes'].data_ptr(), full['nes_binary'].data_ptr(), full['num_neighborhoods_enriched'].data_ptr()])
    ctx.sync()
    return full


def _outputs(bufs, enriched, full, m_total, group, gather, ready=None):
    """Local blocks (NumPy) + the requested full matrices; `full` = whole-matrix tensors after FDR, or None;
    `ready` = full matrices some earlier step already produced (the integer exchange's NES)."""
    dist = _dist()
    ready = ready or {}
    if full is None:
        out = {k: v.cpu().numpy() for k, v in bufs.items()}
        out['num_neighborhoods_enriched'] = enriched.cpu().numpy()
        for k in gather:
            if k in bufs:                    # (the hypergeometric path has no pvalues_neg / ns: safe.py:556-608 leaves them unset)
                out['full_' + k] = (ready[k] if k in ready else gather_columns(bufs[k], m_total, group)).cpu().numpy()
        return out
    c0, c1 = column_shards(m_total, dist.get_world_size(group))[dist.get_rank(group)]
    out = {k: (full[k][:, c0:c1] if k in full else v).cpu().numpy() for k, v in bufs.items()}
    out['num_neighborhoods_enriched'] = full['num_neighborhoods_enriched'][c0:c1].cpu().numpy()
    for k in gather:
        if k in bufs:
            out['full_' + k] = (full[k] if k in full else gather_columns(bufs[k], m_total, group)).cpu().numpy()
    return out


def _call_permutations(ctx, n, flags, num_permutations, random_seed, unseeded, alone, group):
    """The permutation tables of one (collective) call.  Seeded: NumPy's legacy stream, drawn once per node when the ranks
    share one (ensure_shared_stream).  Unseeded: generated on every rank's device from the agreed value -- `random_seed` is
    then rank 0's entropy draw (reduce_flags_and_stats / agree_on_seed), or None in a single process."""
    from . import backend as be
    if unseeded and be.device_stream_enabled() and int(np.count_nonzero(flags)) <= 65535:
        return be.Permutations(ctx, n, flags, num_permutations, None, device_key=random_seed)
    shared = (not alone) and ensure_shared_stream(ctx, group)        # one draw thread per node, not per rank
    if unseeded and random_seed is not None:
        random_seed = int(random_seed) & 0xFFFFFFFF                  # (the agreed entropy value is wider than an MT19937 seed)
    return be.Permutations(ctx, n, flags, num_permutations, random_seed, shared=shared)


RANDOMIZATION_OUTPUTS = ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary')
HYPERGEOM_OUTPUTS = ('pvalues_pos', 'nes', 'nes_binary')


def randomization_step(ctx, nbr, attr, m_total, num_permutations, random_seed, bufs, enriched,
                       neighborhood_score_type='sum', attribute_sign='both', enrichment_threshold=0.05, group=None,
                       table=None, flags=None, exchange=True, timing=None, unseeded=None, overlap=None):
    """One rank's share of compute_pvalues_by_randomization (safe.py:474-554) on DEVICE-resident inputs and
    outputs, plus the path's exchange steps -- the function bench.py times and the host-level drivers below
    call, so what is measured is what runs.

    attr: this rank's column block (backend.Attributes).  bufs: dict name -> f64 [N, M_r] device tensor for
    RANDOMIZATION_OUTPUTS; enriched: f64 [M_r].  flags: whole-matrix row flags when the caller already
    exchanged them (then `random_seed` must already be the agreed seed), else they are exchanged here together
    with the seed.  exchange=True all-gathers the result (integer counters when the kernel left them, f64 NES
    blocks otherwise) and returns the full [N, m_total] NES device tensor; a tuple of names from COUNTER_OUTPUTS returns
    {name: full tensor} for all of them from the same single exchange; False returns None (D2H-only runs: every rank
    keeps its block).  Without a process group it is the plain single-GPU step.  `timing`: a dict that
    receives this rank's host-stream / kernel / exchange times of the step (bench.py).
    random_seed=None (or unseeded=True with `random_seed` = the value the ranks agreed on): an UNSEEDED run -- the tables are
    generated on every rank's device from the agreed value (backend.Permutations, device stream): no host stream, no sharing.
    overlap: True = the ranks agreed (stats['agree'] of the caller's own reduce_flags_and_stats) that every block leaves bit-sliced
    counters: the exchange then travels in column chunks behind the last launches (ChunkedExchange); None = settled here when the
    step runs the head collective itself, off when the caller passed `flags` (SAFE_HIP_XCHG_OVERLAP=0 switches it off)."""
    import os
    import time
    from . import backend as be
    alone = not _dist().is_initialized()               # a single process: the same step without the two exchanges
    if unseeded is None:
        unseeded = random_seed is None
    names = () if not exchange else ('nes',) if exchange is True else tuple(exchange)
    # the overlapped exchange (ChunkedExchange) is settled in the head collective: every rank says whether ITS block will run the
    # bit-sliced kernel (safe_randomization_plan); it needs that collective, so a caller that exchanged the flags itself says
    # what was agreed there with overlap=True
    chunks, cols = (0, 0)
    if not alone and names and os.environ.get('SAFE_HIP_XCHG_OVERLAP', '1') != '0' and bufs[names[0]].is_cuda:
        chunks, cols = exchange_chunk_grid(m_total, _dist().get_world_size(group))
    if flags is None:
        stats = attr.stats()                           # (the dispatch rule's inputs: part of every compute_pvalues pass)
        flags = attr.row_flags()
        # (round 6: the row flags in a pass of their own FIRST, so that the draw chain starts ~60 us earlier while the statistics
        # are computed, measured 3.22 against 3.15 ms per step -- a second sweep of the matrix and a second host wait, and the
        # launching thread still sits in the statistics call before it can enqueue the first stage; not kept)
        if not alone:
            mine_ok = chunks >= 1 and be.randomization_plan(ctx, nbr, attr, int(num_permutations), neighborhood_score_type) == 0
            flags, stats = reduce_flags_and_stats(flags, stats, group, random_seed, agree=1 if mine_ok else 0)   # (cf. _block_agrees)
            random_seed = stats['random_seed']
            if overlap is None:
                overlap = stats['agree'] == 1
    if not alone:
        attr.set_row_flags(flags)                      # indx_vals of the FULL matrix (safe_extras.py:51)
    perms = _call_permutations(ctx, attr.n, flags, int(num_permutations), random_seed, unseeded, alone, group)
    if timing is not None:
        timing['stream_role'] = perms.timing()['role']
    xc = None
    if overlap and chunks >= 1:
        xc = ChunkedExchange(ctx, nbr, bufs, names, m_total, num_permutations, attribute_sign, enrichment_threshold, group, table,
                             chunks, cols)
    try:
        if xc is not None:
            xc.arm()
        be.randomization(ctx, nbr, attr, perms, neighborhood_score_type, attribute_sign, enrichment_threshold,
                         [bufs[k].data_ptr() for k in RANDOMIZATION_OUTPUTS] + [enriched.data_ptr()], table=table)
        t_x = time.perf_counter()
        if timing is not None:
            timing.update(perms.timing())
            name, k_ms, launches = ctx.last_kernel()
            timing.update(kernel=name, gpu_kernel_ms=k_ms * max(int(launches), 1), gpu_kernel_busy_ms=ctx.last_kernel_busy_ms())
        if not exchange:
            return None
        if alone and exchange is True:
            return bufs['nes']
        if alone:
            return {k: bufs[k] for k in names}
        report = {} if timing is not None else None
        if xc is not None:
            full = xc.finish(report)
            if timing is not None:
                # what the step still waited for after its kernels, and the whole window from the first chunk's copy-out on
                timing['exchange_window_ms'] = 1e3 * (time.perf_counter() - xc.t_launched)
        else:
            full = gather_outputs(ctx, nbr, bufs, names, m_total, int(num_permutations), attribute_sign, enrichment_threshold, group,
                                  table=table, report=report)
        if timing is not None:
            timing['exchange_ms'] = 1e3 * (time.perf_counter() - t_x)
            timing['exchange'] = report
        return full['nes'] if exchange is True else full
    finally:
        if xc is not None:
            xc.disarm()
        perms.close()


def _alloc_outputs(ctx, n, mloc, names):
    import torch
    dev = torch.device('cuda', ctx.device)
    return ({k: torch.empty((n, mloc), dtype=torch.float64, device=dev) for k in names},
            torch.empty((mloc,), dtype=torch.float64, device=dev))


def _block_agrees(ctx, nbr, attr, m_total, num_permutations, neighborhood_score_type, group):
    """This rank's word for the head collective: 1 = its block will leave bit-sliced counters and the run is wide enough for the
    chunked exchange (randomization_step's `overlap`)."""
    import os
    from . import backend as be
    if os.environ.get('SAFE_HIP_XCHG_OVERLAP', '1') == '0':
        return 0
    return 1 if be.randomization_plan(ctx, nbr, attr, int(num_permutations), neighborhood_score_type) == 0 else 0


def _randomization_host(ctx, nbr, attr, m_total, num_permutations, random_seed, flags, neighborhood_score_type,
                        attribute_sign, enrichment_threshold, group, gather, multiple_testing, unseeded=False, overlap=False):
    import torch
    bufs, enriched = _alloc_outputs(ctx, attr.n, attr.m, RANDOMIZATION_OUTPUTS)
    torch.cuda.current_stream().synchronize()
    # every requested matrix the counters determine comes out of ONE integer exchange; with multiple_testing the adjusted
    # p-values are whole-matrix quantities and the unadjusted ones travel the same way first (_fdr_whole_matrix)
    wanted = tuple(k for k in COUNTER_OUTPUTS if k in gather) if not multiple_testing else ('pvalues_neg', 'pvalues_pos')
    ready = randomization_step(ctx, nbr, attr, m_total, num_permutations, random_seed, bufs, enriched,
                               neighborhood_score_type, attribute_sign, enrichment_threshold, group, flags=flags,
                               exchange=wanted if wanted else False, unseeded=unseeded, overlap=overlap)
    ctx.sync()
    full = _fdr_whole_matrix(ctx, bufs, m_total, num_permutations, attribute_sign, enrichment_threshold,
                             group, ready=ready) if multiple_testing else None
    return _outputs(bufs, enriched, full, m_total, group, gather, ready=None if multiple_testing else ready)


def sharded_randomization(ctx, nbr, local_attr_host, m_total, num_permutations, random_seed,
                          neighborhood_score_type='sum', attribute_sign='both', enrichment_threshold=0.05,
                          group=None, gather=('nes',), multiple_testing=False):
    """compute_pvalues_by_randomization for this rank's column block + the two exchange steps.
    `local_attr_host`: this rank's [N, M_r] block of node2attribute (NumPy).  Returns a dict
    with the local blocks (NumPy) and, for every name in `gather`, the all-gathered full matrix
    under 'full_<name>' ('nes' travels as integer counters when the kernel keeps them, see
    gather_nes).  random_seed=None: rank 0's entropy seed is used by every rank (the result is
    then one unseeded run of the whole matrix, not `world` unrelated ones).  Needs a HIP device
    on every rank (no CPU fallback)."""
    from . import backend as be
    attr = be.Attributes.from_host(ctx, local_attr_host)
    try:
        stats0 = attr.stats()
        flags, stats = reduce_flags_and_stats(attr.row_flags(), stats0, group, random_seed,
                                              agree=_block_agrees(ctx, nbr, attr, m_total, num_permutations, neighborhood_score_type, group))
        return _randomization_host(ctx, nbr, attr, m_total, num_permutations, stats['random_seed'], flags,
                                   neighborhood_score_type, attribute_sign, enrichment_threshold, group, gather,
                                   multiple_testing, unseeded=random_seed is None, overlap=stats['agree'] == 1)
    finally:
        attr.close()


def permutation_split_randomization(ctx, nbr, attr_host, num_permutations, random_seed, neighborhood_score_type='sum',
                                    attribute_sign='both', enrichment_threshold=0.05, group=None, multiple_testing=False):
    """compute_pvalues_by_randomization split along the PERMUTATION axis -- for matrices with fewer attributes than
    ranks (one scatter-plot column, BASELINE configs[0]), where column shards leave GPUs idle.  The reference sketches
    this split for its worker processes (safe.py:489-519: num_permutations / processes each, counts added up); there
    every worker reseeds identically and repeats the others' permutations.  Here EVERY rank passes the whole [N, M]
    matrix and draws the one cumulative stream of the call (safe_extras.py:46-58 -- a permutation depends on all earlier
    ones, so the host draws cannot be skipped), tests only its own range of it (safe_perms_slice) and the integer
    counts are summed over the ranks (one all-reduce: the path's real exchange step).  The sum equals the
    single-process counts, so p-values / NES / nes_binary are those of the unsplit call, on every rank.
    Returns the same dict as sharded_randomization with every 'full_<name>' equal to the local matrix."""
    import torch
    from . import backend as be
    dist = _dist()
    alone = not dist.is_initialized()
    world = 1 if alone else dist.get_world_size(group)
    rank = 0 if alone else dist.get_rank(group)
    total = int(num_permutations)
    attr = be.Attributes.from_host(ctx, attr_host)
    try:
        seed = random_seed if alone else agree_on_seed(random_seed, group)
        n, m = attr.n, attr.m
        dev = torch.device('cuda', ctx.device)
        ns, neg, pos = (torch.zeros((n, m), dtype=torch.float64, device=dev) for _ in range(3))
        torch.cuda.current_stream().synchronize()
        p0, p1 = column_shards(total, world)[rank]
        whole = _call_permutations(ctx, n, attr.row_flags(), total, seed, random_seed is None, alone, group)
        try:
            if p1 > p0:
                mine = whole.slice(p0, p1)
                try:
                    be.permtest_counts(ctx, nbr, attr, mine, neighborhood_score_type, ns.data_ptr(), neg.data_ptr(), pos.data_ptr())
                finally:
                    mine.close()
            else:                                           # more ranks than permutations: nothing to test, the scores only
                be.score(ctx, nbr, attr, neighborhood_score_type, ns.data_ptr())
            ctx.sync()
        finally:
            whole.close()
        counts = torch.stack([neg, pos]).to(torch.int64)   # exact: counts are whole numbers <= num_permutations
        if not alone:
            if dist.get_backend(group) == 'gloo':
                host = counts.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                counts = host.to(dev)
            else:
                dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        neg, pos = counts[0].to(torch.float64).contiguous(), counts[1].to(torch.float64).contiguous()
        bufs, enriched = _alloc_outputs(ctx, n, m, ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'))
        torch.cuda.current_stream().synchronize()
        be.outputs_from_counts(ctx, n, m, total, attribute_sign, enrichment_threshold, neg.data_ptr(), pos.data_ptr(), ns.data_ptr(),
                               [bufs[k].data_ptr() for k in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary')] + [enriched.data_ptr()])
        if multiple_testing:                               # whole rows are here already (safe.py:536-542)
            be.fdr_adjust(ctx, n, m, total, attribute_sign, enrichment_threshold,
                          [bufs['pvalues_neg'].data_ptr(), bufs['pvalues_pos'].data_ptr(), bufs['nes'].data_ptr(),
                           bufs['nes_binary'].data_ptr(), enriched.data_ptr()])
        ctx.sync()
        out = {k: v.cpu().numpy() for k, v in bufs.items()}
        out['ns'] = ns.cpu().numpy()
        out['num_neighborhoods_enriched'] = enriched.cpu().numpy()
        for k in RANDOMIZATION_OUTPUTS:
            out['full_' + k] = out[k]
        out['stats'] = {'random_seed': seed, 'permutation_range': (p0, p1)}
        out['how'] = 'randomization'
        return out
    finally:
        attr.close()


def _hypergeom_host(ctx, nbr, attr, m_total, flags, enrichment_threshold, group, gather, multiple_testing, attribute_sign):
    import torch
    from . import backend as be
    attr.stats()
    attr.set_row_flags(flags)
    bufs, enriched = _alloc_outputs(ctx, attr.n, attr.m, HYPERGEOM_OUTPUTS)
    torch.cuda.current_stream().synchronize()
    be.hypergeom(ctx, nbr, attr, enrichment_threshold, [bufs[k].data_ptr() for k in HYPERGEOM_OUTPUTS] + [enriched.data_ptr()])
    ctx.sync()
    full = _fdr_whole_matrix(ctx, bufs, m_total, 0, attribute_sign, enrichment_threshold, group) if multiple_testing else None
    return _outputs(bufs, enriched, full, m_total, group, gather)


def sharded_hypergeom(ctx, nbr, local_attr_host, m_total, global_flags, enrichment_threshold=0.05, group=None,
                      gather=('nes',), multiple_testing=False, attribute_sign='both'):
    """compute_pvalues_by_hypergeom for this rank's column block.  The population N
    (safe.py:574-578) and the neighborhood sizes count the rows holding a value in ANY column of
    the full matrix, so the flags of the whole matrix are installed before the kernels run."""
    from . import backend as be
    attr = be.Attributes.from_host(ctx, local_attr_host)
    try:
        return _hypergeom_host(ctx, nbr, attr, m_total, global_flags, enrichment_threshold, group, gather,
                               multiple_testing, attribute_sign)
    finally:
        attr.close()


def sharded_compute_pvalues(ctx, nbr, local_attr_host, m_total, enrichment_type='auto', num_permutations=1000,
                            random_seed=None, neighborhood_score_type='sum', attribute_sign='both',
                            enrichment_threshold=0.05, group=None, gather=('nes',), multiple_testing=False):
    """SAFE.compute_pvalues (safe.py:432-472) over attribute shards with WHOLE-MATRIX semantics:
    the 'auto' rule (safe.py:461-463) looks at every column of the full matrix (one all-gather of
    the per-shard statistics), so all ranks take the same branch and the result equals the
    single-process call on the unsplit matrix -- unlike the reference's CLI (safe.py:1321-1361),
    whose worker processes each decide for their own chunk.  multiple_testing=True adjusts every row
    across ALL attributes as the unsplit call does (the p-value blocks are gathered first).  One upload
    of the block, one exchange of flags / statistics / seed, then the branch."""
    from . import backend as be
    attr = be.Attributes.from_host(ctx, local_attr_host)
    try:
        stats0 = attr.stats()
        flags, stats = reduce_flags_and_stats(attr.row_flags(), stats0, group, random_seed,
                                              agree=_block_agrees(ctx, nbr, attr, m_total, num_permutations, neighborhood_score_type, group))
        if (enrichment_type == 'hypergeometric') or (enrichment_type == 'auto' and stats['n_other'] == 0):
            out = _hypergeom_host(ctx, nbr, attr, m_total, flags, enrichment_threshold, group, gather,
                                  multiple_testing, attribute_sign)
            out['how'] = 'hypergeometric'
        else:
            out = _randomization_host(ctx, nbr, attr, m_total, num_permutations, stats['random_seed'], flags,
                                      neighborhood_score_type, attribute_sign, enrichment_threshold, group, gather,
                                      multiple_testing, unseeded=random_seed is None, overlap=stats['agree'] == 1)
            out['how'] = 'randomization'
    finally:
        attr.close()
    out['stats'] = stats
    return out
