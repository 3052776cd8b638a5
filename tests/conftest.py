import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_nbr():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, 'neighborhoods.npz')))


@pytest.fixture(scope='session')
def golden_enr():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, 'enrichment.npz')))


@pytest.fixture(scope='session')
def golden_rng():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, 'rng_kat.npz')))


@pytest.fixture(scope='session')
def golden_fdr():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, 'fdr.npz')))


@pytest.fixture(scope='session')
def golden_big():
    """tests/golden/big.npz decoded to the reference's own arrays (N = 1200, 300 permutations)."""
    import numpy as np
    g = dict(np.load(os.path.join(GOLDEN, 'big.npz')))
    n = g['xy'].shape[0]
    out = {k: g[k] for k in ('xy', 'edge_u', 'edge_v', 'edge_length', 'b_q')}
    out['A'] = np.unpackbits(g['A_bits'], axis=1)[:, :n].astype(np.int64)
    assert np.array_equal(out['A'].sum(axis=1), g['A_row_counts'])
    b = np.unpackbits(g['b_bin'], axis=0)[:n].astype(np.float32)
    b[g['b_bin_nan_rows']] = np.nan
    out['b_bin'] = np.asfortranarray(b)
    for tag in ('bin', 'q_sum', 'q_z'):
        nperm, seed = (int(v) for v in g[tag + '_meta'])
        out[tag + '_nperm'], out[tag + '_seed'] = nperm, seed
        for side in ('neg', 'pos'):
            c = g[tag + '_counts_' + side].astype(np.float64)
            out[tag + '_pvalues_' + side] = np.where(c < 0, np.nan, c / float(nperm))
        out[tag + '_nes'] = g[tag + '_nes_values'][g[tag + '_nes_codes']]
        out[tag + '_nes_binary'] = g[tag + '_nes_binary'].astype(np.float64)
        out[tag + '_num_enriched'] = g[tag + '_num_enriched']
        out[tag + '_ns'] = g[tag + '_ns'].astype(np.float64)
    return out
