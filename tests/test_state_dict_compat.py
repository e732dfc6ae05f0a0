"""CPU: checkpoint wire format -- the product modules expose exactly the reference's state_dict keys, shapes and
parameter order, so reference `iter_N.pth` files load and vice versa.  Constructors only: no kernel runs."""
import pytest
import torch
from segtran_amd import engine
from util import golden_json

KEYS = golden_json('state_dict_keys')


@pytest.mark.parametrize('cfg', ['cfg1', 'cfg2', 'cfg4', 'cfg5'])
def test_state_dict_keys_and_shapes(cfg):
    net = engine.build_model(cfg, 'cpu', synth=False)
    sd = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert sd == KEYS[cfg]
    assert list(sd) == list(KEYS[cfg]), 'state_dict order differs'
    if cfg + '_params' in KEYS:
        assert [k for k, _ in net.named_parameters()] == KEYS[cfg + '_params']


def test_tied_qk_is_one_parameter_and_attribute_surface():
    net = engine.build_model('cfg1', 'cpu', synth=False)
    tl = net.voxel_fusion.translayers[0]
    for att in (tl.in_ator_trans, tl.ator_out_trans):
        assert att.key.weight is att.query.weight and att.key.bias is att.query.bias          # N2
        assert hasattr(att.out_trans, 'first_linear')
    assert net.backbone.endpoint_blk_indices == [2, 6, 10, 22] and len(net.backbone._blocks) == 32
    assert net.num_vis_layers == 3 and hasattr(net, 'layers_attn_scores') and hasattr(net, 'feature_maps')
    low = [n for n, _ in net.named_parameters() if 'backbone' in n]
    assert len(low) > 400                                                                   # low-decay group selector


def test_invalid_inputs_raise_like_the_reference_breakpoints():
    net = engine.build_model('cfg1', 'cpu', synth=False)
    with pytest.raises(ValueError, match='divisible by 8'):
        net(torch.zeros(1, 3, 60, 64))
