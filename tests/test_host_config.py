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
