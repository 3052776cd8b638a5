"""One rank of the multi-process GPU parity test (tests/test_gpu_multirank.py starts `world` of these as
fresh child processes).  Every rank builds the same seeded inputs, runs the product's sharded driver
(safepy_amd.sharding.sharded_compute_pvalues) on its np.array_split column block, and checks

    all-gathered result == the single-process SAFE.compute_pvalues on the unsplit matrix == the oracle

usage: multirank_worker.py RANK WORLD PORT BACKEND OUTDIR [STREAM]
BACKEND nccl = RCCL, one GPU per rank (needs >= WORLD devices); gloo = every rank on device 0 with the
exchange staged through the host (runs on a one-GPU box; same kernels, same integer-counter exchange).
STREAM shared (default) = one permutation stream per node: LOCAL_RANK / LOCAL_WORLD_SIZE are set as a launcher would and
local rank 0 draws for everyone (safe_perms_create_shared); own = every rank draws the whole stream itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cases(rng, n):
    """(name, attribute matrix, kwargs of compute_pvalues).  Column counts are odd so the split is uneven."""
    binary = (rng.uniform(size=(n, 131)) < 0.04).astype(np.float32)
    binary[rng.choice(n, 12, replace=False)] = np.nan
    binary[5, :] = np.nan
    binary[5, 130] = 1.0                        # a row whose only value sits in the LAST rank's block
    quant = rng.normal(size=(n, 45))
    quant[rng.choice(n, 10, replace=False)] = np.nan
    quant[rng.uniform(size=quant.shape) < 0.01] = np.nan
    quant[7, :] = np.nan
    quant[7, 44] = 0.25
    return [('binary-randomization', np.asfortranarray(binary), dict(how='randomization', num_permutations=60)),
            ('binary-auto-hypergeometric', binary.astype(np.float64), dict(how='auto')),
            ('quantitative-sum', quant, dict(how='auto', num_permutations=50)),
            ('quantitative-zscore', quant.astype(np.float32), dict(how='auto', num_permutations=30, neighborhood_score_type='z-score')),
            ('quantitative-fdr', quant, dict(how='auto', num_permutations=40, multiple_testing=True))]


def main():
    rank, world, port, backend, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    stream = sys.argv[6] if len(sys.argv) > 6 else 'shared'
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if stream == 'shared':
        os.environ.update(LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    else:
        os.environ['SAFE_HIP_SHARED_STREAM'] = '0'
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')     # as bench.py / run_batch.py run
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    device = rank if backend == 'nccl' else 0
    torch.cuda.set_device(device)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import safepy_amd
        from safepy_amd import sharding
        from oracle import safe_oracle as orc               # the checker
        rng = np.random.default_rng(21)
        n = 700
        xy = rng.uniform(size=(n, 2))
        sf = safepy_amd.SAFE(verbose=False, device=device)
        sf.graph = safepy_amd.LayoutGraph(xy)
        sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.08)
        a = sf.neighborhoods
        assert np.array_equal(a, orc.neighborhoods_euclidean(xy, 0.08))
        ctx, nbr = sf._ctx(), sf._device_neighborhoods()
        for name, b, kw in cases(rng, n):
            m = b.shape[1]
            c0, c1 = sharding.column_shards(m, world)[rank]
            score = kw.get('neighborhood_score_type', 'sum')
            fdr = kw.get('multiple_testing', False)
            out = sharding.sharded_compute_pvalues(
                ctx, nbr, np.ascontiguousarray(b[:, c0:c1]), m, enrichment_type=kw['how'],
                num_permutations=kw.get('num_permutations', 1000), random_seed=9, neighborhood_score_type=score,
                gather=('nes', 'nes_binary', 'pvalues_pos', 'pvalues_neg'), multiple_testing=fdr)
            # ---- single process, unsplit matrix
            sf.random_seed = 9
            sf.load_attributes(attribute_file=b.copy())
            sf.compute_pvalues(**dict(kw, neighborhood_score_type=score, multiple_testing=fdr))     # (kwargs persist on the object, like the reference's)
            assert out['how'] == ('hypergeometric' if name == 'binary-auto-hypergeometric' else 'randomization'), name
            # every full matrix the reference leaves on the instance -- for the counter kernels all four come out of ONE
            # exchange of packed integers (sharding.gather_outputs)
            for key in ('nes', 'nes_binary', 'pvalues_pos') + (('pvalues_neg',) if out['how'] == 'randomization' else ()):
                assert np.array_equal(out['full_' + key], getattr(sf, key), equal_nan=True), (name, key, rank)
                assert np.array_equal(out[key], getattr(sf, key)[:, c0:c1], equal_nan=True), (name, key, rank)
            assert np.array_equal(out['num_neighborhoods_enriched'],
                                  sf.attributes['num_neighborhoods_enriched'].values[c0:c1]), name
            if out['how'] == 'randomization':
                assert np.array_equal(out['pvalues_neg'], sf.pvalues_neg[:, c0:c1], equal_nan=True), name
                np.testing.assert_allclose(out['ns'], sf.ns[:, c0:c1], rtol=1e-12, atol=0, equal_nan=True)
            # ---- the oracle on the unsplit matrix
            want = orc.compute_pvalues(a, b.astype(b.dtype).copy(), enrichment_type=kw['how'],
                                       num_permutations=kw.get('num_permutations', 1000), random_seed=9,
                                       neighborhood_score_type=score, multiple_testing=fdr)
            assert np.array_equal(out['full_nes_binary'], want['nes_binary'], equal_nan=True), name
            if out['how'] == 'randomization':
                assert np.array_equal(out['full_pvalues_pos'], want['pvalues_pos'], equal_nan=True), name
                if fdr:        # NES of adjusted p-values: the device's log10, not the k/P table of NumPy values
                    np.testing.assert_allclose(out['full_nes'], want['nes'], rtol=1e-12, atol=1e-12, equal_nan=True)
                else:
                    assert np.array_equal(out['full_nes'], want['nes'], equal_nan=True), name
            else:
                np.testing.assert_allclose(out['full_pvalues_pos'], want['pvalues_pos'], rtol=1e-6, atol=1e-300)
                np.testing.assert_allclose(out['full_nes'], want['nes'], rtol=1e-6, atol=1e-9)

        # ---- a network beyond the blocked kernel's 16-bit offsets (N = 9000: the sixteen-wave pre-permuted form leaves the packed
        # counters the exchange carries): sharded == single process
        n9 = 9000
        xy9 = np.random.default_rng(5).uniform(size=(n9, 2))
        sf9 = safepy_amd.SAFE(verbose=False, device=device)
        sf9.graph = safepy_amd.LayoutGraph(xy9)
        sf9.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.02)
        b9 = (np.random.default_rng(6).uniform(size=(n9, 131)) < 0.05).astype(np.float32)
        b9[np.random.default_rng(7).choice(n9, 50, replace=False)] = np.nan
        c0, c1 = sharding.column_shards(131, world)[rank]
        out = sharding.sharded_compute_pvalues(sf9._ctx(), sf9._device_neighborhoods(), np.ascontiguousarray(b9[:, c0:c1]), 131,
                                               enrichment_type='randomization', num_permutations=30, random_seed=4,
                                               gather=('nes', 'nes_binary', 'pvalues_pos', 'pvalues_neg'))
        assert sf9._ctx().last_kernel()[0] == 'k_permtest_bits_pre', sf9._ctx().last_kernel()
        sf9.random_seed = 4
        sf9.load_attributes(attribute_file=b9.copy())
        sf9.compute_pvalues(how='randomization', num_permutations=30)
        for key in ('nes', 'nes_binary', 'pvalues_pos', 'pvalues_neg'):
            assert np.array_equal(out['full_' + key], getattr(sf9, key), equal_nan=True), ('N=9000', key, rank)
        del sf9

        # ---- the chunked exchange (sharding.ChunkedExchange): settled in the head collective, one slab by default; a block wide
        # enough to be cut into column chunks; the column-chunked tail of the launches (SAFE_HIP_XCHG_TAIL) behind which they travel
        from safepy_amd import backend as be
        wide = (rng.uniform(size=(n, 256 * world + 77)) < 0.03).astype(np.float32)
        wide[rng.choice(n, 9, replace=False)] = np.nan
        mw = wide.shape[1]
        c0, c1 = sharding.column_shards(mw, world)[rank]
        want_wide = {}
        for P in (300, 60, 1100):
            sf.random_seed = 9
            sf.load_attributes(attribute_file=wide.copy())
            sf.compute_pvalues(how='randomization', num_permutations=P, neighborhood_score_type='sum', multiple_testing=False)
            want_wide[P] = {k: getattr(sf, k).copy() for k in ('nes', 'nes_binary', 'pvalues_pos', 'pvalues_neg')}
            want_wide[P]['enriched'] = sf.attributes['num_neighborhoods_enriched'].values.copy()

        def xc_env(chunks, tail):
            for key, val in (('SAFE_HIP_XCHG_CHUNKS', chunks), ('SAFE_HIP_XCHG_TAIL', tail)):
                if val is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = val

        for P, chunks, tail, want_made, what in (
                (300, None, None, 0, 'one slab (the default)'),
                (300, '4', None, 0, 'column chunks after the kernels'),
                (300, '4', '0.4', 2, 'column-chunked tail of the launches'),
                (300, '2', '1e-9', 2, 'the last stage as the tail'),
                (60, '4', '0.4', 0, 'too few stages for a tail: the armed grid served from the finished counters'),
                (1100, None, None, 0, 'more than 1023 permutations: u32 counter pairs'),
                (1100, '4', None, 0, 'u32 counter pairs in column chunks')):
            xc_env(chunks, tail)
            attr = be.Attributes.from_host(ctx, np.ascontiguousarray(wide[:, c0:c1]))
            bufs, enriched = sharding._alloc_outputs(ctx, n, c1 - c0, sharding.RANDOMIZATION_OUTPUTS)
            torch.cuda.current_stream().synchronize()
            t = {}
            full = sharding.randomization_step(ctx, nbr, attr, mw, P, 9, bufs, enriched, 'sum', 'both', 0.05,
                                               exchange=('nes', 'nes_binary', 'pvalues_pos', 'pvalues_neg'), timing=t)
            want_chunks = 1 if chunks is None else 2          # (the widest block has five word groups: two chunks at most)
            assert t['exchange'].get('chunks') == want_chunks and 'agreed before the kernels' in t['exchange']['form'], (what, t['exchange'])
            # P <= 1023: the counter pair travels as 10 + 10 bits, two outputs in five bytes -- 0.625 of the u32 slab
            n_pad_x, cols_x = 64 * (-(-n // 64)), sharding.exchange_chunk_grid(mw, world)[1]
            slab_u32 = cols_x * n_pad_x + sharding.ChunkedExchange.HEADER
            want_words = (cols_x * (n_pad_x // 8 * 5) + sharding.ChunkedExchange.HEADER) if P <= 1023 else slab_u32
            assert t['exchange']['form'].startswith('packed 10 + 10 bit' if P <= 1023 else 'packed u32'), (what, t['exchange'])
            assert t['exchange']['bytes_received'] == 4 * want_words * want_chunks * (world - 1), (what, t['exchange'])
            assert P > 1023 or want_words <= 0.65 * slab_u32
            made = be.packed_chunk_info(ctx)[0]
            assert made == want_made, (what, made)
            for k, v in full.items():
                assert np.array_equal(v.cpu().numpy(), want_wide[P][k], equal_nan=True), (what, k, rank)
            assert np.array_equal(bufs['nes'].cpu().numpy(), want_wide[P]['nes'][:, c0:c1], equal_nan=True), what
            assert np.array_equal(enriched.cpu().numpy(), want_wide[P]['enriched'][c0:c1]), what
            attr.close()
        # the product's driver takes the same path, and a repeated call reuses the armed state cleanly
        xc_env('4', '0.4')
        out = sharding.sharded_compute_pvalues(ctx, nbr, np.ascontiguousarray(wide[:, c0:c1]), mw, enrichment_type='randomization',
                                               num_permutations=300, random_seed=9, gather=('nes', 'pvalues_neg'))
        assert np.array_equal(out['full_nes'], want_wide[300]['nes'], equal_nan=True)
        assert np.array_equal(out['full_pvalues_neg'], want_wide[300]['pvalues_neg'], equal_nan=True)
        # a rank whose call leaves no bit-sliced counters although it said it would (here: forced onto the f64 kernels after the
        # agreement): it still takes part in every chunk's collective, flags its slabs, and ALL ranks fall back to f64 blocks
        if world > 1:
            for chunks, tail in (('4', '0.4'), (None, None)):
                xc_env(chunks, tail)
                attr = be.Attributes.from_host(ctx, np.ascontiguousarray(wide[:, c0:c1]))
                bufs, enriched = sharding._alloc_outputs(ctx, n, c1 - c0, sharding.RANDOMIZATION_OUTPUTS)
                torch.cuda.current_stream().synchronize()
                real_plan = be.randomization_plan
                if rank == world - 1:
                    os.environ['SAFE_HIP_FORCE_PATH'] = 'gather'
                    be.randomization_plan = lambda *a, **k: 0
                t = {}
                try:
                    full = sharding.randomization_step(ctx, nbr, attr, mw, 300, 9, bufs, enriched, 'sum', 'both', 0.05,
                                                       exchange=('nes', 'pvalues_pos'), timing=t)
                finally:
                    be.randomization_plan = real_plan
                    os.environ.pop('SAFE_HIP_FORCE_PATH', None)
                assert t['exchange']['form'].startswith('f64 blocks'), t['exchange']
                for k, v in full.items():
                    assert np.array_equal(v.cpu().numpy(), want_wide[300][k], equal_nan=True), ('fallback', k, rank)
                attr.close()
        xc_env(None, None)
        # SAFE_HIP_XCHG_OVERLAP=0: the exchange that agrees after the kernels (gather_outputs) -- the same matrices
        os.environ['SAFE_HIP_XCHG_OVERLAP'] = '0'
        out = sharding.sharded_compute_pvalues(ctx, nbr, np.ascontiguousarray(wide[:, c0:c1]), mw, enrichment_type='randomization',
                                               num_permutations=300, random_seed=9, gather=('nes',))
        assert np.array_equal(out['full_nes'], want_wide[300]['nes'], equal_nan=True)
        del os.environ['SAFE_HIP_XCHG_OVERLAP']

        # who drew: with a shared stream only local rank 0 runs a draw thread, the others fetched every chunk from its ring
        assert (ctx.shared_stream is not None) == (stream == 'shared' and world > 1), ctx.shared_stream
        flags = np.ones(n, dtype=np.uint8)
        flags[::7] = 0
        perms = be.Permutations(ctx, n, flags, 300, 5, shared=True)           # collective: three pipeline stages
        table = perms.read()
        role = perms.timing()['role']
        perms.close()
        assert role == ('own' if ctx.shared_stream is None else 'producer' if rank == 0 else 'consumer'), role
        mine = be.Permutations(ctx, n, flags, 300, 5)                         # this rank's own stream: the same tables
        assert np.array_equal(table, mine.read())
        mine.close()
        assert np.array_equal(table[0][flags == 0], np.flatnonzero(flags == 0))     # rows without a value never move

        # random_seed=None: ONE unseeded run of the whole matrix -- every rank generates the SAME tables on its device from the
        # value rank 0 drew, so the all-gathered matrix is a consistent run: the full matrices are identical on all ranks and
        # equal the single-process unseeded run keyed with that value
        name, b, kw = cases(np.random.default_rng(21), n)[0]
        m = b.shape[1]
        c0, c1 = sharding.column_shards(m, world)[rank]
        out = sharding.sharded_compute_pvalues(ctx, nbr, np.ascontiguousarray(b[:, c0:c1]), m, enrichment_type='randomization',
                                               num_permutations=40, random_seed=None, gather=('nes', 'pvalues_pos'))
        seed = out['stats']['random_seed']
        seeds = [None] * world
        dist.all_gather_object(seeds, seed)
        assert len(set(seeds)) == 1, seeds
        sf.random_seed = None
        sf.device_stream_key = seed
        sf.load_attributes(attribute_file=b.copy())
        sf.compute_pvalues(how='randomization', num_permutations=40, neighborhood_score_type='sum', multiple_testing=False)
        assert np.array_equal(out['full_nes'], sf.nes, equal_nan=True)
        assert np.array_equal(out['full_pvalues_pos'], sf.pvalues_pos, equal_nan=True)
        sf.device_stream_key = None
        # the same with the device stream switched off: the NumPy-compatible stream from the agreed seed, as in round 2
        os.environ['SAFE_HIP_DEVICE_STREAM'] = '0'
        out = sharding.sharded_compute_pvalues(ctx, nbr, np.ascontiguousarray(b[:, c0:c1]), m, enrichment_type='randomization',
                                               num_permutations=40, random_seed=None, gather=('nes',))
        sf.random_seed = out['stats']['random_seed'] & 0xFFFFFFFF       # (the agreed value has 63 bits: its low 32 seed MT19937)
        sf.load_attributes(attribute_file=b.copy())
        sf.compute_pvalues(how='randomization', num_permutations=40, neighborhood_score_type='sum', multiple_testing=False)
        assert np.array_equal(out['full_nes'], sf.nes, equal_nan=True)
        del os.environ['SAFE_HIP_DEVICE_STREAM']
        # permutation-axis split (fewer attributes than ranks -- BASELINE configs[0] is ONE column): every rank passes the
        # whole matrix, tests its range of the one stream, the counts are all-reduced; the result on every rank is the
        # single-process result
        one = rng.normal(size=(n, 1))
        one[rng.choice(n, 200, replace=False)] = np.nan
        one[3, 0] = 0.0
        two = (rng.uniform(size=(n, 2)) < 0.05).astype(np.float64)
        for name, b, kw in (('one-column-sum', one, dict(num_permutations=53)),
                            ('one-column-zscore', one, dict(num_permutations=31, neighborhood_score_type='z-score')),
                            ('two-binary-columns', two, dict(num_permutations=40)),
                            ('one-column-fdr', one, dict(num_permutations=30, multiple_testing=True)),
                            ('fewer-permutations-than-ranks', one, dict(num_permutations=world - 1))):
            score = kw.get('neighborhood_score_type', 'sum')
            fdr = kw.get('multiple_testing', False)
            out = sharding.permutation_split_randomization(ctx, nbr, b, kw['num_permutations'], 9, neighborhood_score_type=score,
                                                           multiple_testing=fdr)
            p0, p1 = out['stats']['permutation_range']
            assert (p0, p1) == sharding.column_shards(kw['num_permutations'], world)[rank], name
            if kw['num_permutations'] >= 10:            # (SAFE itself refuses fewer, like the reference: safe.py validate_config)
                sf.random_seed = 9
                sf.load_attributes(attribute_file=b.copy())
                sf.compute_pvalues(how='randomization', num_permutations=kw['num_permutations'], neighborhood_score_type=score,
                                   multiple_testing=fdr)
                for key in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
                    assert np.array_equal(out[key], getattr(sf, key), equal_nan=True), (name, key, rank)
                assert np.array_equal(out['num_neighborhoods_enriched'], sf.attributes['num_neighborhoods_enriched'].values), name
            if not fdr:
                want = orc.compute_pvalues(a, b.copy(), enrichment_type='randomization', num_permutations=kw['num_permutations'],
                                           random_seed=9, neighborhood_score_type=score)
                for key in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
                    assert np.array_equal(out[key], want[key], equal_nan=True), (name, key, rank)
        # permutation-axis split of an UNSEEDED call: every rank generates the whole table on its device from the agreed value
        out = sharding.permutation_split_randomization(ctx, nbr, one, 60, None)
        sf.random_seed = None
        sf.device_stream_key = out['stats']['random_seed']
        sf.load_attributes(attribute_file=one.copy())
        sf.compute_pvalues(how='randomization', num_permutations=60, neighborhood_score_type='sum', multiple_testing=False)
        for key in ('pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
            assert np.array_equal(out[key], getattr(sf, key), equal_nan=True), ('unseeded split', key, rank)
        sf.device_stream_key = None
        if backend == 'nccl':
            # the exchange through the C ABI's own RCCL communicator (safe_comm_* / safe_allgather_cols): the id travels
            # over the process group here; a non-torch host would use a file or MPI
            from safepy_amd import backend as be
            box = [be.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = be.Comm(ctx, world, rank, box[0])
            nbytes = 3 << 20
            mine, everyone = ctx.alloc(nbytes), ctx.alloc(world * nbytes)
            mine.upload(np.full(nbytes, 17 + rank, dtype=np.uint8))
            comm.allgather(mine.ptr, nbytes, everyone.ptr)
            ctx.sync()
            got = everyone.download((world, nbytes), dtype=np.uint8)
            assert all((got[r] == 17 + r).all() for r in range(world))
            comm.close()
        open(os.path.join(outdir, 'ok%d' % rank), 'w').write('ok')
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
