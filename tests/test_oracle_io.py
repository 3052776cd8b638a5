"""Oracle restatements of the io-side functions against the golden vectors
captured from the real reference (tests/golden/io.npz, made by tests/golden/make_golden.py io)."""
import os

import numpy as np
import pytest

from oracle import safe_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def io_golden():
    return dict(np.load(os.path.join(HERE, 'golden', 'io.npz')))


def _write(tmp_path, name, data):
    p = os.path.join(str(tmp_path), name)
    with open(p, 'wb') as f:
        f.write(data.tobytes())
    return p


def test_pseudo_network_edges(io_golden):
    g = io_golden
    xy = np.stack([g['scatter_x'], g['scatter_y']], axis=1)
    assert np.array_equal(orc.pseudo_network_edges(xy, 0.07), g['scatter_pseudo_edges'])
    assert np.array_equal(g['scatter_pseudo_weights'], [1.0])


def test_scatter_neighborhoods(io_golden):
    g = io_golden
    xy = np.stack([g['scatter_x'], g['scatter_y']], axis=1)
    assert np.array_equal(orc.neighborhoods_euclidean(xy, 0.07), g['scatter_neighborhoods'])


def test_weighted_edge_lengths(io_golden):
    g = io_golden
    got = orc.weighted_edge_lengths(g['wl_xy'], g['wl_edge_u'], g['wl_edge_v'], g['wl_weight'])
    assert np.array_equal(got, g['wl_length'], equal_nan=True)
    assert np.isnan(g['wl_length']).sum() == (g['wl_weight'] == 0).sum() > 0
    assert g['wl_length'][-1] == 0.0                      # the self loop


@pytest.mark.parametrize('tag,ext', [('ra_bin', '.txt'), ('ra_q', '.txt.gz'), ('ra_f32', '.txt')])
def test_read_attributes_files(io_golden, tmp_path, tag, ext):
    g = io_golden
    path = _write(tmp_path, tag + ext, g[tag + '_file'])
    names, table = orc.parse_attribute_text(path)
    order, mat = orc.align_attributes(table, list(g['ra_node_order']))
    want = g[tag + '_matrix']
    assert mat.dtype == want.dtype
    assert np.array_equal(mat, want, equal_nan=True)
    assert names == list(g[tag + '_names'])


def test_read_attributes_dataframe(io_golden):
    import pandas as pd
    g = io_golden
    frame = pd.DataFrame(g['ra_df_values'], index=list(g['ra_df_index']), columns=list('abcdef'))
    np.random.seed(3)
    order, mat = orc.align_attributes(frame, list(g['ra_node_order']), fill_value=0, mask_duplicates=True)
    assert np.array_equal(mat, g['ra_df_matrix'], equal_nan=True)
    assert np.isnan(mat).any() and (mat == 0).any()
    order, mat = orc.align_attributes(frame)
    assert np.array_equal(mat, g['ra_df_noorder_matrix'])
    assert order == list(g['ra_df_noorder_order'])
