"""The reference's only published timing is its Example 3 notebook (examples/Example_3_Scatterplot_annotation.ipynb:
73, 104, 147-153): a 1586-node `.scatter` file -> load_network -> define_neighborhoods('euclidean', 0.06) -> ONE quantitative
attribute from a DataFrame -> compute_pvalues(num_permutations=10000).  The same call sequence on a surrogate of that shape
(safe-data is not available offline), through the drop-in class, against the oracle EXACTLY: the attribute carries 10
fractional bits, so every neighborhood sum is exact in f64 and the empirical p-values have one right answer."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import safe_oracle as orc            # noqa: E402  (checker only)


def test_example3_shape_against_the_oracle(tmp_path):
    import safepy_amd
    from safepy_amd import workloads
    assert safepy_amd.device_count() >= 1, 'no HIP device: the GPU tests must run on the MI355X box'
    path = os.path.join(str(tmp_path), 'surrogate_UMAP_1586.scatter')
    keys, xy, att = workloads.example3_scatter(path)
    nperm, seed = 10000, 7

    sf = safepy_amd.SAFE(verbose=False)
    sf.random_seed = seed
    sf.load_network(network_file=path, node_key_attribute='key')
    sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.06)
    sf.load_attributes(attribute_file=att)
    sf.compute_pvalues(num_permutations=nperm)

    a = orc.neighborhoods_euclidean(xy, 0.06)
    assert np.array_equal(sf.neighborhoods, a)
    b = att.to_numpy(dtype=np.float64)
    assert np.array_equal(sf.node2attribute, b, equal_nan=True)
    want = orc.compute_pvalues(a, b.copy(), enrichment_type='auto', num_permutations=nperm, random_seed=seed)
    for key in ('ns', 'pvalues_neg', 'pvalues_pos', 'nes', 'nes_binary'):
        assert np.array_equal(np.asarray(getattr(sf, key)), want[key], equal_nan=True), key
    assert np.array_equal(sf.attributes['num_neighborhoods_enriched'].values, want['num_neighborhoods_enriched'])
    assert 0 < sf.nes_binary.sum() < a.shape[0]          # the surrogate has enriched and unenriched neighborhoods
