"""N > 1 plumbing on CPU: world_size-2 gloo process group (no GPU).  Covers the two exchange
steps of the attribute-sharded path (global row flags / statistics, final all-gather) and the
shard arithmetic; the oracle plays the role of the per-rank compute so the assertion is
'sharded == unsharded' on the same seeded inputs."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.distributed as dist            # noqa: E402
import torch.multiprocessing as mp          # noqa: E402

from oracle import safe_oracle as orc       # noqa: E402  (checker only)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from safepy_amd import sharding
        rng = np.random.default_rng(5)                      # same data on every rank
        n, m, nperm, seed = 120, 17, 12, 4
        xy = rng.uniform(size=(n, 2))
        a = orc.neighborhoods_euclidean(xy, 0.2)
        b = rng.normal(size=(n, m))
        b[rng.choice(n, 9, replace=False)] = np.nan
        b[:, 3] = np.nan
        # a row whose only values sit in the LAST rank's columns: per-shard flags differ
        b[7, :] = np.nan
        b[7, m - 1] = 1.5
        shards = sharding.column_shards(m, world)
        assert shards == [(0, 9), (9, 17)]
        c0, c1 = shards[rank]
        local = b[:, c0:c1]

        # exchange step 1: whole-matrix row flags and statistics
        local_flags = (~np.isnan(local)).any(axis=1).astype(np.uint8)
        flags = sharding.reduce_row_flags(local_flags)
        assert np.array_equal(flags, (~np.isnan(b)).any(axis=1).astype(np.uint8))
        if rank == 0:
            assert local_flags[7] == 0 and flags[7] == 1
        st = sharding.reduce_stats({'n_other': int((~np.isnan(local) & ~np.isin(local, [0, 1])).sum()),
                                    'n_non_integer': int((~np.isnan(local) & (local != np.floor(local))).sum()),
                                    'max_nan_col': int(np.isnan(local).sum(axis=0).max())})
        assert st['n_other'] == int((~np.isnan(b) & ~np.isin(b, [0, 1])).sum())
        assert st['max_nan_col'] == n
        # the same two reductions as ONE collective (what bench.py and the sharded driver use)
        flags1, st1 = sharding.reduce_flags_and_stats(local_flags, {
            'n_other': int((~np.isnan(local) & ~np.isin(local, [0, 1])).sum()),
            'n_non_integer': int((~np.isnan(local) & (local != np.floor(local))).sum()),
            'max_nan_col': int(np.isnan(local).sum(axis=0).max())})
        assert np.array_equal(flags1, flags) and flags1.dtype == np.uint8
        assert {k: v for k, v in st1.items() if k not in ('random_seed', 'agree')} == st and st1['random_seed'] == 0
        assert st1['agree'] == 1
        # the word the ranks settle before the kernels (the overlapped exchange): the smallest one wins
        _, st1b = sharding.reduce_flags_and_stats(local_flags, st, agree=1 if rank == 0 else 0)
        assert st1b['agree'] == 0
        assert sharding.exchange_chunk_grid(8 * 3971, 8) == (1, 4032) and sharding.exchange_chunk_grid(131, 2) == (1, 128)
        os.environ['SAFE_HIP_XCHG_CHUNKS'] = '4'
        try:
            assert sharding.exchange_chunk_grid(8 * 3971, 8) == (4, 1024) and sharding.exchange_chunk_grid(131, 2) == (1, 128)
            chunks, cols = sharding.exchange_chunk_grid(2 * 256 + 77, 2)
            assert chunks == 2 and cols % 64 == 0 and chunks * cols >= 295
        finally:
            del os.environ['SAFE_HIP_XCHG_CHUNKS']
        # one permutation stream for the whole matrix (safe_extras.py:46, 58): a given seed is kept, an unset one
        # (random_seed=None) becomes rank 0's draw on EVERY rank
        _, st2 = sharding.reduce_flags_and_stats(local_flags, st, random_seed=1234 + rank)
        assert st2['random_seed'] == 1234                       # rank 0's value wins
        _, st3 = sharding.reduce_flags_and_stats(local_flags, st, random_seed=None)
        seeds = [None, None]
        dist.all_gather_object(seeds, st3['random_seed'])
        assert seeds[0] == seeds[1] and 0 <= seeds[0] < 2 ** 63       # 63 bits of rank 0's entropy (the device stream's key)
        agreed = sharding.agree_on_seed(None)
        dist.all_gather_object(seeds, agreed)
        assert seeds[0] == seeds[1] and sharding.agree_on_seed(77) == 77

        # per-rank compute (oracle stand-in for the HIP kernels) with the GLOBAL permutation stream
        table = orc.permutation_index_table(np.where(flags[:, None] > 0, 0.0, np.nan) * np.ones((n, 1)), nperm, seed)
        obs = orc.compute_neighborhood_score(a, local, 'sum')
        cn = np.zeros(obs.shape)
        cp = np.zeros(obs.shape)
        for k in range(nperm):
            s = orc.compute_neighborhood_score(a, local[table[k]], 'sum')
            cn += s <= obs
            cp += s >= obs

        # exchange step 2: all-gather of the result blocks
        full_cn = sharding.gather_columns(torch.from_numpy(cn), m).numpy()
        full_cp = sharding.gather_columns(torch.from_numpy(cp), m).numpy()
        want_cn, want_cp = orc.run_permutations(a, b, 'sum', nperm, seed)
        assert np.array_equal(full_cn, want_cn) and np.array_equal(full_cp, want_cp)
        # gather_nes without device counters (gloo): agrees on the fallback and moves the f64 blocks
        full_again = sharding.gather_nes(None, None, torch.from_numpy(cn), m, nperm, 'both').numpy()
        assert np.array_equal(full_again, want_cn)
        # gather_outputs: every requested matrix in one call; without device counters every rank agrees on the f64 fallback and
        # reports it (the device path, one exchange of packed integers for all of them, is tests/test_gpu_multirank.py)
        report = {}
        both = sharding.gather_outputs(None, None, {'pvalues_neg': torch.from_numpy(cn / nperm), 'pvalues_pos': torch.from_numpy(cp / nperm)},
                                       ('pvalues_neg', 'pvalues_pos'), m, nperm, 'both', 0.05, report=report)
        assert np.array_equal(both['pvalues_neg'].numpy(), want_cn / nperm) and np.array_equal(both['pvalues_pos'].numpy(), want_cp / nperm)
        assert report['form'] == 'f64 blocks' and report['bytes_received'] > 0
        # the node-shared permutation stream is a device feature: without a context nothing is shared, and the decision is
        # remembered per context so that no rank can enter the setup's broadcast alone
        assert sharding.ensure_shared_stream(None) is False
        # multiple_testing=True under sharding: p-value blocks gathered, full rows adjusted on every rank (the oracle stands in
        # for safe_fdr_adjust), local blocks = slices of the adjusted matrix; untouched outputs stay the rank's own
        p_local = cp / nperm
        p_full = sharding.gather_columns(torch.from_numpy(p_local), m).numpy()
        adj = np.apply_along_axis(orc.fdrcorrection, 1, p_full)
        assert np.array_equal(adj, np.apply_along_axis(orc.fdrcorrection, 1, want_cp / nperm))
        full = {'pvalues_pos': torch.from_numpy(adj), 'nes': torch.from_numpy(-np.log10(adj)),
                'nes_binary': torch.from_numpy((adj < 0.05).astype(np.float64)),
                'num_neighborhoods_enriched': torch.from_numpy((adj < 0.05).sum(axis=0).astype(np.float64))}
        bufs = {'ns': torch.from_numpy(obs), 'pvalues_pos': torch.from_numpy(p_local), 'nes': torch.from_numpy(p_local * 0),
                'nes_binary': torch.from_numpy(p_local * 0)}
        out = sharding._outputs(bufs, torch.zeros(c1 - c0, dtype=torch.float64), full, m, None, ('nes', 'pvalues_pos'))
        assert np.array_equal(out['pvalues_pos'], adj[:, c0:c1]) and np.array_equal(out['full_pvalues_pos'], adj)
        assert np.array_equal(out['nes'], -np.log10(adj)[:, c0:c1]) and np.array_equal(out['ns'], obs, equal_nan=True)
        assert np.array_equal(out['num_neighborhoods_enriched'], (adj < 0.05).sum(axis=0)[c0:c1])
        plain = sharding._outputs(bufs, torch.ones(c1 - c0, dtype=torch.float64), None, m, None, ('pvalues_pos',))
        assert np.array_equal(plain['full_pvalues_pos'], p_full) and np.array_equal(plain['pvalues_pos'], p_local)
        open(os.path.join(tmpdir, 'ok%d' % rank), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_equals_unsharded(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ('ok%d' % r)).exists() for r in range(world))


def test_column_shards_match_array_split():
    from safepy_amd import sharding
    for m in (1, 7, 8, 4373, 50000):
        for world in (1, 2, 3, 8):
            want = [(int(c[0]), int(c[-1]) + 1) if len(c) else None for c in np.array_split(np.arange(m), world)]
            got = sharding.column_shards(m, world)
            for w, g in zip(want, got):
                if w is None:
                    assert g[0] == g[1]
                else:
                    assert w == g
            assert got[0][0] == 0 and got[-1][1] == m
