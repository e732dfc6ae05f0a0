"""CPU: flag surface of the trainer mirrors (no kernels run: argument handling and defaults only)."""
import argparse
import pytest
from segtran_amd import train_common as tc


def _parse(argv, dim):
    p = tc.common_flags(argparse.ArgumentParser(), dim)
    return tc.finalize_args(p.parse_args(argv), dim)


def test_reference_flag_names_and_segtran_defaults():
    a = _parse(['--task', 'fundus', '--net', 'segtran', '--translayers', '3', '--layercompress', '1,1,2,2', '--bs', '6', '--noqkbias'], 2)
    assert (a.num_translayers, a.translayer_compress_ratios, a.batch_size, a.qk_have_bias) == (3, [1, 1, 2, 2], 6, False)
    assert (a.lr, a.decay, a.grad_clip, a.dropout_prob, a.num_modes, a.num_attractors) == (2e-4, 1e-4, 0.1, 0.2, 4, 256)
    b = _parse(['--task', 'brats', '--translayers', '1', '--attractors', '1024', '--bs', '4'], 3)
    assert (b.num_attractors, b.translayer_compress_ratios, b.batch_size) == (1024, [1, 1], 4)


@pytest.mark.parametrize('flag', ['--mince', '--multihead', '--attnconsist', '--squeezeuseffn', '--inbn', '--outdrop'])
def test_out_of_scope_features_are_rejected_loudly(flag):
    with pytest.raises(SystemExit):
        _parse([flag], 2)


def test_other_networks_are_rejected():
    with pytest.raises(SystemExit):
        _parse(['--net', 'unet'], 2)
