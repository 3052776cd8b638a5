"""Seeded runs reproduce NumPy's legacy stream (np.random.seed + np.random.permutation, safepy/safe_extras.py:46-58).  One host
thread runs the MT19937 / masked-rejection chain and ships the accepted swap targets (2 bytes each); the swaps themselves are
replayed ON THE DEVICE (k_replay_targets: one wave per permutation, 64 steps at a time with the steps that share a location
settled in lane order) and composed by the scan kernels.  These tests pin the composed tables bit for bit against NumPy
itself, over sizes that exercise every branch of the replay: 0 / 1 / 2 movable rows, batches with many colliding targets
(small k), more steps than one batch, more chunks than staging buffers, rows that never move, the 32-bit / global-memory
form beyond 65535 movable rows, and a handle whose buffers are reused with another set of movable rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def amd():
    import safepy_amd
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    return safepy_amd


@pytest.fixture(scope='module')
def ctx(amd):
    return amd.Context.default(0)


def numpy_tables(n, flags, nperm, seed):
    """np.random.seed(seed); nperm x [perm = np.random.permutation(indx_vals); cur[indx_vals] = cur[perm]] -- the reference's loop."""
    np.random.seed(seed)
    movable = np.flatnonzero(flags)
    cur = np.arange(n)
    out = np.empty((nperm, n), dtype=np.int64)
    for q in range(nperm):
        perm = np.random.permutation(movable)
        cur[movable] = cur[perm]
        out[q] = cur
    return out


@pytest.mark.parametrize('n,k_fixed,nperm,seed', [
    (1, 0, 5, 0), (1, 1, 5, 0), (2, 0, 40, 1), (3, 1, 40, 2), (5, 4, 6, 3), (4, 4, 3, 4),
    (17, 0, 300, 5), (64, 0, 130, 6), (65, 1, 130, 7), (66, 0, 130, 8), (129, 3, 200, 9), (300, 17, 500, 10),
    (1000, 0, 385, 11), (3971, 182, 300, 0), (3971, 182, 1000, 12345), (3971, 182, 10000, 99), (2500, 0, 2555, 4294967295), (8193, 1, 40, 13), (20000, 1000, 150, 14),
    (32769, 0, 12, 15), (65535, 0, 6, 16), (65536, 0, 5, 17)])
def test_tables_equal_numpy(amd, ctx, n, k_fixed, nperm, seed):
    from safepy_amd import backend as be
    rng = np.random.default_rng(n + 7)
    flags = np.ones(n, dtype=np.uint8)
    flags[rng.choice(n, k_fixed, replace=False)] = 0
    perms = be.Permutations(ctx, n, flags, nperm, seed)
    got = perms.read().astype(np.int64)
    assert perms.timing()['role'] == 'own'
    perms.close()
    np.testing.assert_array_equal(got, numpy_tables(n, flags, nperm, seed))
    assert (got[:, flags == 0] == np.flatnonzero(flags == 0)).all()            # rows without a value never move


def test_beyond_16_bit_positions(amd, ctx):
    """k > 65535 movable rows: 32-bit targets on the wire, replay on a global-memory array."""
    from safepy_amd import backend as be
    n = 70001
    flags = np.ones(n, dtype=np.uint8)
    flags[::9] = 0
    perms = be.Permutations(ctx, n, flags, 3, 21)
    got = perms.read().astype(np.int64)
    perms.close()
    np.testing.assert_array_equal(got, numpy_tables(n, flags, 3, 21))


def test_reused_handle_with_other_rows_and_the_oracle(amd, ctx):
    """The context keeps the buffers of the last destroyed handle of a shape: a second call with the same (n, count) but other
    movable rows must not see anything of the first; and the oracle's own restatement of the stream agrees with NumPy."""
    from safepy_amd import backend as be
    n, nperm = 777, 260
    rng = np.random.default_rng(3)
    for trial in range(3):
        flags = (rng.uniform(size=n) < (0.9, 0.5, 0.02)[trial]).astype(np.uint8)
        perms = be.Permutations(ctx, n, flags, nperm, 40 + trial)
        got = perms.read().astype(np.int64)
        perms.close()
        np.testing.assert_array_equal(got, numpy_tables(n, flags, nperm, 40 + trial))
    b = np.where(flags[:, None] == 1, 1.0, np.nan) * np.ones((n, 2))
    np.testing.assert_array_equal(orc.permutation_index_table(b, 20, 42), numpy_tables(n, flags, 20, 42))


def test_partial_reads_follow_the_pipeline(amd, ctx):
    """Rows are produced chunk by chunk (32, 96, 128 ... permutations); reading a prefix must not need the rest."""
    from safepy_amd import backend as be
    n, nperm = 500, 1000
    flags = np.ones(n, dtype=np.uint8)
    want = numpy_tables(n, flags, nperm, 77)
    perms = be.Permutations(ctx, n, flags, nperm, 77)
    np.testing.assert_array_equal(perms.read(0, 10).astype(np.int64), want[:10])
    np.testing.assert_array_equal(perms.read(120, 300).astype(np.int64), want[120:300])
    np.testing.assert_array_equal(perms.read().astype(np.int64), want)
    perms.close()


@pytest.mark.parametrize('n,k_fixed,nperm,seed', [(3971, 182, 1000, 7), (300, 17, 700, 3), (20000, 1000, 150, 14)])
def test_twin_chain_gives_the_same_tables(amd, ctx, monkeypatch, n, k_fixed, nperm, seed):
    """SAFE_HIP_DRAW_TWIN=1 (opt-in): two host threads draw the same chain, whichever finishes a pipeline chunk first publishes
    it -- the tables must not depend on who won which chunk."""
    from safepy_amd import backend as be
    rng = np.random.default_rng(n + 7)
    flags = np.ones(n, dtype=np.uint8)
    flags[rng.choice(n, k_fixed, replace=False)] = 0
    monkeypatch.setenv('SAFE_HIP_DRAW_TWIN', '1')
    want = numpy_tables(n, flags, nperm, seed)
    for _ in range(3):                                          # (who wins a chunk changes from run to run)
        perms = be.Permutations(ctx, n, flags, nperm, seed)
        got = perms.read().astype(np.int64)
        t = perms.timing()
        perms.close()
        assert t['twin_chain'] and t['chunks'] >= 1
        np.testing.assert_array_equal(got, want)
    monkeypatch.setenv('SAFE_HIP_DRAW_TWIN', '0')
    perms = be.Permutations(ctx, n, flags, nperm, seed)
    got = perms.read().astype(np.int64)
    assert not perms.timing()['twin_chain']
    perms.close()
    np.testing.assert_array_equal(got, want)
