"""Host logic that needs no device: the INI reader of the SAFE mirror (safepy/safe.py:116-188) --
every key the reference reads reaches the instance, with the reference's defaults and types."""
import os

import pytest


def _write(tmp_path, text):
    p = tmp_path / 'user.ini'
    p.write_text(text)
    return str(p)


def test_defaults_match_the_reference_defaults():
    import safepy_amd
    sf = safepy_amd.SAFE(verbose=False)
    # safepy/safe_default.ini:1-24 through safe.py:149-184
    assert sf.path_to_safe_data is None
    assert sf.path_to_network_file == 'networks/Costanzo_Science_2016.gpickle'
    assert sf.path_to_attribute_file == 'attributes/hoepfner_movva_2014_doxorubucin.txt'
    assert sf.attribute_sign == 'both'
    assert sf.background == 'attribute_file'
    assert sf.node_distance_metric == 'shortpath_weighted_layout'
    assert sf.neighborhood_radius == 0.1 and isinstance(sf.neighborhood_radius, float)
    assert sf.neighborhood_radius_type == 'diameter'
    assert sf.random_seed is None
    assert sf.attribute_unimodality_metric == 'connectivity'
    assert sf.attribute_distance_metric == 'jaccard'
    assert sf.attribute_distance_threshold == 0.75 and isinstance(sf.attribute_distance_threshold, float)
    assert os.path.isdir(sf.output_dir)


def test_user_ini_reaches_every_attribute(tmp_path):
    import safepy_amd
    ini = _write(tmp_path, '\n'.join([
        '[Input files]',
        'safe_data = /data/safe/',
        'networkfile = nets/my.gpickle',
        'annotationfile = attrs/my.txt',
        'annotationsign = highest\t# OPTIONS: highest, lowest, both',
        '[Analysis parameters]',
        'background = network',
        'nodeDistanceType = euclidean',
        'neighborhoodRadius = 0.25',
        'neighborhoodRadiusType = absolute',
        'randomSeed = 7',
        'unimodalityType = none',
        'groupDistanceType = hamming',
        'groupDistanceThreshold = 0.5',
    ]))
    sf = safepy_amd.SAFE(path_to_ini_file=ini, verbose=False)
    assert sf.path_to_safe_data == '/data/safe/'
    assert sf.path_to_network_file == '/data/safe/nets/my.gpickle'
    assert sf.path_to_attribute_file == '/data/safe/attrs/my.txt'
    assert sf.attribute_sign == 'highest'
    assert sf.background == 'network'
    assert sf.node_distance_metric == 'euclidean'
    assert sf.neighborhood_radius == 0.25
    assert sf.neighborhood_radius_type == 'absolute'
    assert sf.random_seed == 7
    assert sf.attribute_unimodality_metric == 'none'
    assert sf.attribute_distance_metric == 'hamming'
    assert sf.attribute_distance_threshold == 0.5
    assert sf.output_dir == str(tmp_path)


def test_constructor_data_path_wins_and_must_end_with_a_slash(tmp_path):
    import safepy_amd
    ini = _write(tmp_path, '[Input files]\nsafe_data = /from/ini/\n')
    sf = safepy_amd.SAFE(path_to_ini_file=ini, path_to_safe_data='/from/ctor/', verbose=False)
    assert sf.path_to_network_file.startswith('/from/ctor/')
    with pytest.raises(AssertionError):
        safepy_amd.SAFE(path_to_safe_data='/no/trailing/slash', verbose=False)


def test_invalid_option_restores_the_default_and_raises(tmp_path):
    import safepy_amd
    ini = _write(tmp_path, '[Analysis parameters]\nnodeDistanceType = manhattan\n')
    with pytest.raises(ValueError):
        safepy_amd.SAFE(path_to_ini_file=ini, verbose=False)
    sf = safepy_amd.SAFE(verbose=False)
    sf.node_distance_metric = 'manhattan'
    with pytest.raises(ValueError):
        sf.validate_config()
    assert sf.node_distance_metric == 'shortpath_weighted_layout'      # safe.py:205-209


# ---------------------------------------------------------------------------------------------------------------------
# the lazy result attributes and their recycled host arrays (safepy_amd/safe.py _LazyArray), with a stand-in for the device buffer
# ---------------------------------------------------------------------------------------------------------------------
class _FakeBuffer:
    def __init__(self, value):
        self.value, self.freed, self.into = value, False, None

    def download(self, shape, out=None):
        import numpy as np
        arr = np.empty(shape) if out is None else out
        self.into = out
        arr[...] = self.value
        return arr

    def free(self):
        self.freed = True


def test_lazy_result_arrays_are_recycled_only_when_nobody_holds_them():
    import numpy as np
    import pickle
    import safepy_amd
    from safepy_amd import safe as S
    sf = safepy_amd.SAFE(verbose=False)
    b1 = _FakeBuffer(1.0)
    sf.nes = S._DeviceResult(b1, (5, 3))
    first = sf.nes                                             # first read: copied to the host, the device buffer released
    assert b1.freed and b1.into is None and (first == 1.0).all()
    assert sf.nes is first                                     # later reads: the same host array
    # the caller still holds `first`: the next result must not be written into it
    b2 = _FakeBuffer(2.0)
    sf.nes = S._DeviceResult(b2, (5, 3))
    second = sf.nes
    assert second is not first and b2.into is None and (first == 1.0).all() and (second == 2.0).all()
    # nobody holds the second result any more: the third one is written into the same memory
    addr = second.ctypes.data
    del second
    b3 = _FakeBuffer(3.0)
    sf.nes = S._DeviceResult(b3, (5, 3))
    third = sf.nes
    assert third.ctypes.data == addr and b3.into is third and (third == 3.0).all()
    # another shape: a fresh array; a value the caller assigned is never recycled
    del third
    sf.nes = S._DeviceResult(_FakeBuffer(4.0), (2, 2))
    assert sf.nes.shape == (2, 2) and sf.nes.ctypes.data != addr
    mine = np.zeros((2, 2))
    sf.nes = mine
    b5 = _FakeBuffer(5.0)
    sf.nes = S._DeviceResult(b5, (2, 2))
    assert sf.nes is not mine and (mine == 0.0).all()
    # the bookkeeping is not part of a pickled object
    state = sf.__getstate__()
    assert not [k for k in state if k.startswith('_spare_') or k.startswith('_made_')]
    clone = pickle.loads(pickle.dumps(sf))
    assert (clone.nes == 5.0).all()
