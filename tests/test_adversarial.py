"""the few-shot / adversarial training step (reference train2d.py:1147-1186, 1259-1286, 1314-1326) mirrored as host control flow
(segtran_amd/engine.py: attach_adversarial, domain_adversarial_loss, AdversarialTrainStep) over the HIP discriminator / BertAdam.

Fixture tests/golden/adversarial.npz: the reference's OWN loop statements executed on a toy network (make_golden.py case_adversarial)."""
import os
import numpy as np
import pytest
import torch

from segtran_amd import engine, functional as SF
from segtran_amd.synth import synth_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'adversarial.npz')


class ToyNet(torch.nn.Module):
    def __init__(self, Cf=8, nc=3):
        super().__init__()
        self.body = torch.nn.Conv2d(3, Cf, 3, padding=1)
        self.head = torch.nn.Conv2d(Cf, nc, 1)
        self.feature_maps, self.discriminator, self.recon = [], None, None

    def forward(self, x):
        f = torch.tanh(self.body(x))
        self.feature_maps = [f]
        return self.head(f)


def _close(a, b, tol, what, scale=None):
    b = torch.as_tensor(np.asarray(b)).to(a.device)
    scale = scale or max(float(b.abs().max()), 1e-12)
    err = float((a - b).abs().max())
    assert err <= tol * scale, '%s: err %.3e of scale %.3e' % (what, err, scale)


@pytest.mark.parametrize('mode', ['feat', 'mask'])
@pytest.mark.parametrize('variant', ['revgrad', 'adda', 'adda0'])
def test_domain_adversarial_loss_vs_reference_loop(backend, mode, variant):
    """revgrad: gradient reversal in front of the discriminator, one optimizer.  adda / adda0 (make_golden.ADDA_LRS): the discriminator steps with its
    own optimizer inside the loss computation and the generator is trained on inverted labels -- at learning rate 1e-3 the stepped weights and the
    loss values are compared, at learning rate 0 every gradient of the second backward pass (see make_golden.py for why two runs)."""
    from segtran_amd.networks.discriminator import Discriminator
    from segtran_amd.optimization import BertAdam
    g = np.load(GOLD)
    dev = backend.dev
    adda = variant != 'revgrad'
    lr = {'revgrad': 0.0, 'adda': 1e-3, 'adda0': 0.0}[variant]
    tag = '%s_%s' % (mode, variant)
    net = ToyNet().to(dev)
    net.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('toy:')})
    dis = Discriminator(8 if mode == 'feat' else 3, num_classes=1, do_revgrad=not adda, num_base_chan=8).to(dev)
    dis.load_state_dict({k: v.to(dev) for k, v in synth_state_dict({k: tuple(v.shape) for k, v in dis.state_dict().items()}).items()})
    dis.train()
    before = {k: p.detach().clone() for k, p in dis.named_parameters()}
    net.discriminator = dis
    image, source = torch.from_numpy(g['image']).to(dev), torch.from_numpy(g['source']).to(dev)
    dis_opt = None
    if adda:
        dis_opt = BertAdam([dict(params=list(dis.parameters()), weight_decay=0.0, lr=lr)], lr=lr, warmup=-1, t_total=-1, weight_decay=0.0)
        dis_opt.release_flat_grads()
    out = SF.interp_linear(net(image), (64, 64))
    domain_loss, _ = engine.domain_adversarial_loss(net, mode, image, source, torch.sigmoid(out) if mode == 'mask' else None, (64, 64), adda, dis_opt)
    _close(domain_loss.detach(), g[tag + ':domain_loss'], 2e-5, 'domain loss')
    loss = 1.0 * ((1 - 0.5) * out.square().mean() + 0.5 * out.abs().mean()) + 0.002 * domain_loss       # the fixture's stand-in supervised terms
    _close(loss.detach(), g[tag + ':loss'], 2e-5, 'total loss')
    if variant == 'adda':
        for k, p in dis.named_parameters():
            ref = torch.from_numpy(g[tag + ':disparam:' + k]).to(dev)
            moved, ref_moved = p.detach() - before[k].to(dev), ref - before[k].to(dev)          # the step itself (~3e-3 per element)
            assert float(ref_moved.abs().max()) > 1e-4 and float((moved - ref_moved).abs().max()) <= 0.02 * float(ref_moved.abs().max()), k
        return
    for p in net.parameters():              # as the fixture does (net.zero_grad() covers the attached discriminator): the ADDA discriminator step's
        p.grad = None                        # retain_graph pass left gradients on generator and discriminator alike
    loss.backward()
    for k, p in net.named_parameters():
        if k.startswith('discriminator.'):
            continue
        _close(p.grad, g[tag + ':toygrad:' + k], 1e-3, 'generator gradient ' + k)
    gscale = max(float(np.abs(g[tag + ':disgrad:' + k]).max()) for k, _ in dis.named_parameters())     # the global gradient scale, as in the model-level tests
    for k, p in dis.named_parameters():
        _close(p.grad, g[tag + ':disgrad:' + k], 1e-3, 'discriminator gradient ' + k, gscale)


def test_adversarial_train_step_runs_the_whole_recipe(backend):
    """AdversarialTrainStep on a toy network: supervised rows in front, unsupervised target rows behind, the source batch through the same network,
    reconstruction head, both optimizers; the reported parts add up to the loss and the discriminator learns to tell the domains apart."""
    from segtran_amd.optimization import BertAdam
    dev = backend.dev
    torch.manual_seed(0)
    net = ToyNet(Cf=8, nc=3).to(dev)
    engine.attach_adversarial(net, 'feat', num_classes=3, num_feat_dis_in_chan=8, adda=True, recon_w=0.1, device=dev, num_base_chan=8)
    assert net.discriminator is not None and net.recon is not None
    gen = [p for n, p in net.named_parameters() if not n.startswith('discriminator.')]
    opt = BertAdam([dict(params=gen, weight_decay=0.0, lr=1e-3)], lr=1e-3, warmup=-1, t_total=-1, weight_decay=0.0)
    dis_opt = BertAdam([dict(params=list(net.discriminator.parameters()), weight_decay=0.0, lr=5e-3)], lr=5e-3, warmup=-1, t_total=-1, weight_decay=0.0)
    step = engine.AdversarialTrainStep(net, opt, 'fundus', 'feat', adda=True, discriminator_optim=dis_opt, supervised_w=1.0, domain_w=0.002, recon_w=0.1)
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(2, 3, 32, 32, generator=g, device='cpu').to(dev)
    raw = (torch.rand(2, 3, 32, 32, generator=g, device='cpu') > 0.5).float().to(dev) * 255        # fundus mask encoding
    tgt = (torch.randn(2, 3, 32, 32, generator=g, device='cpu') + 1.5).to(dev)
    src = (torch.randn(2, 3, 32, 32, generator=g, device='cpu') - 1.5).to(dev)
    doms = []
    for _ in range(3):
        loss = step(x, raw, tgt, src)
        total = 1.0 * step.parts['supervised'] + 0.002 * step.parts['domain'] + 0.1 * step.parts['recon']
        assert abs(float(loss) - float(total)) < 1e-5
        doms.append(float(step.parts['domain']))
    assert all(d == d for d in doms)
    assert doms[-1] > doms[0]          # parts['domain'] is the INVERTED-label loss (the generator's): it rises as the discriminator gets better


def _toy_step(dev, mode, supervised_w, recon_w=0.0, nc=3):
    from segtran_amd.optimization import BertAdam
    torch.manual_seed(0)
    net = ToyNet(Cf=8, nc=nc).to(dev)
    engine.attach_adversarial(net, mode, num_classes=nc, num_feat_dis_in_chan=8, adda=False, recon_w=recon_w, device=dev, num_base_chan=8)
    opt = BertAdam([dict(params=list(net.parameters()), weight_decay=0.0, lr=1e-3)], lr=1e-3, warmup=-1, t_total=-1, weight_decay=0.0)
    return net, engine.AdversarialTrainStep(net, opt, 'fundus', mode, supervised_w=supervised_w, domain_w=0.002, recon_w=recon_w)


def test_unsupervised_only_step_publishes_stats_and_resamples_to_the_mask_size(backend):
    """ADVICE r04 (train_common.py:310, engine.py:272): `--adv mask --supweight 0` is a supported recipe (train2d.py:1166-1170, 1232-1245: only the
    unsupervised images go through the network, the supervised terms are zeros).  The step must (a) publish a stats tensor laid out like TrainStep's
    (the training loop logs it on every log iteration), (b) still resample the outputs to the MASK size before the sigmoid / discriminator
    (train2d.py:1218-1220, 1274-1277 do so unconditionally) -- the discriminator input is checked through a forward hook."""
    dev = backend.dev
    net, step = _toy_step(dev, 'mask', supervised_w=0.0)
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(2, 3, 16, 16, generator=g, device='cpu').to(dev)
    raw = (torch.rand(2, 3, 32, 32, generator=g, device='cpu') > 0.5).float().to(dev) * 255          # masks at twice the output size
    tgt, src = torch.randn(3, 3, 16, 16, generator=g, device='cpu').to(dev), torch.randn(2, 3, 16, 16, generator=g, device='cpu').to(dev)
    seen = []
    h = net.discriminator.register_forward_hook(lambda m, inp, out: seen.append(tuple(inp[0].shape)))
    loss = step(x, raw, tgt, src)
    h.remove()
    assert seen == [(5, 3, 32, 32)]                          # 2 source + 3 unsupervised target rows (no supervised rows at weight 0), at the mask size
    st = step.stats.detach().cpu()
    assert st.shape == (3 + 3,) and abs(float(st[0]) - float(loss)) < 1e-6 and float(st[1:].abs().sum()) == 0.0
    assert float(step.parts['supervised']) == 0.0 and float(step.parts['domain']) > 0.0
    # with the supervised part on: same layout, the supervised slots are the supervised statistics, slot 0 the TOTAL loss
    net, step = _toy_step(dev, 'mask', supervised_w=1.0)
    loss = step(x, raw, tgt, src)
    st = step.stats.detach().cpu()
    assert st.shape == (6,) and abs(float(st[0]) - float(loss)) < 1e-6 and float(st[1]) > 0.0 and float(st[2]) > 0.0


def test_adversarial_step_rejects_inconsistent_batches(backend):
    dev = backend.dev
    x = torch.zeros(1, 3, 8, 8, device=dev)
    raw = torch.zeros(1, 3, 8, 8, device=dev)
    _, step = _toy_step(dev, 'feat', supervised_w=1.0)
    with pytest.raises(ValueError, match='source batch'):
        step(x, raw, None, x)
    with pytest.raises(ValueError, match='SUPERVISED_W'):
        step(None, None, x, x)
    _, step = _toy_step(dev, None, supervised_w=1.0)
    with pytest.raises(ValueError, match='supervised image batch'):
        step(None, raw, None, None)
