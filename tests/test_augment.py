"""In-step label maps and the 3-D RandomResizedCrop (--randscale) against fixtures produced by the REFERENCE's own functions
(tests/golden/make_golden.py case_augment: datasets2d.py:90-139, 200-223; datasets3d.py:16-40, 611-665).  Runs on the fiber emulator
(CPU) and on the HIP build (-m gpu) through the `backend` fixture; the oracle restatement is checked on the CPU."""
import torch

from segtran_amd import functional as SF
from segtran_amd.dataloaders import datasets3d as D3
from util import golden, assert_close


def test_oracle_label_maps_and_crop_match_reference_fixture():
    from oracle import segtran_oracle as O
    g = golden('augment')
    for ex in (0, 1):
        assert torch.equal(O.fundus_map_mask(g['fundus_in'], bool(ex)), g['fundus_excl%d' % ex].float())
    assert torch.equal(O.polyp_map_mask(g['fundus_in'][:, :1].repeat(1, 3, 1, 1)), g['polyp'].float())
    assert torch.equal(O.brats_map_label(g['brats_in'].long()), g['brats'].float())
    for seed in g['rrc_seeds'].tolist():
        for iso in (1, 0):
            torch.manual_seed(seed)
            v, m = O.random_resized_crop(g['rrc_vol'], g['rrc_mask'].float(), (14, 16, 10), (-0.3, 0.3), bool(iso))
            assert torch.equal(v, g['rrc_v_%d_%d' % (seed, iso)]) and torch.equal(m, g['rrc_m_%d_%d' % (seed, iso)])


def test_label_maps_vs_reference(backend):
    g = golden('augment')
    fm = g['fundus_in'].to(backend.dev)
    assert torch.equal(SF.label_nhot(fm, 'fundus').cpu(), g['fundus_excl0'].float())
    assert torch.equal(SF.label_nhot(fm, 'fundus', exclusive=True).cpu(), g['fundus_excl1'].float())          # --exclusive (datasets2d.py:110-111)
    assert not torch.equal(g['fundus_excl0'], g['fundus_excl1'])
    assert torch.equal(SF.label_nhot(fm[:, :1].repeat(1, 3, 1, 1), 'polyp').cpu(), g['polyp'].float())
    assert torch.equal(SF.label_nhot(g['brats_in'].to(backend.dev), 'brats').cpu(), g['brats'].float())


def test_random_resized_crop_vs_reference(backend):
    """Same seed -> same scale / crop offsets as the reference (the draws come from torch's CPU generator in the reference's order);
    values within fp32 rounding of F.interpolate + F.pad + crop (measured: bit-identical or 1 ulp)."""
    g = golden('augment')
    vol, mask = g['rrc_vol'].to(backend.dev), g['rrc_mask'].float().to(backend.dev)
    for seed in g['rrc_seeds'].tolist():
        for iso in (1, 0):
            torch.manual_seed(seed)
            v, m = D3.RandomResizedCrop(vol, mask, (14, 16, 10), (-0.3, 0.3), isotropic=bool(iso))
            assert v.shape == (1, 2, 14, 16, 10) and m.shape == (1, 4, 14, 16, 10)
            assert_close(v, g['rrc_v_%d_%d' % (seed, iso)], 1e-6, 'volume seed %d' % seed)
            assert_close(m, g['rrc_m_%d_%d' % (seed, iso)], 1e-6, 'mask seed %d' % seed)
            pad_ref = g['rrc_v_%d_%d' % (seed, iso)] == 0
            assert torch.equal((v.cpu() == 0) | ~pad_ref, torch.ones_like(pad_ref)), 'zero padding differs'
    # identity: no rescale, no pad -> the input itself
    v, m = D3.RandomResizedCrop(vol, mask, (14, 16, 10), (0, 0))
    assert torch.equal(v, vol) and torch.equal(m, mask)
