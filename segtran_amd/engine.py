"""Train-step engine shared by train2d.py / train3d.py / bench.py / smoke(): model construction from the
reference's flag set, synthetic batches, and ONE full train step
(forward -> BCE+Dice -> backward -> [gradient all-reduce] -> global clip + BertAdam)."""
from argparse import Namespace
import torch
import torch.nn.functional as F

from . import functional as SF
from . import segx
from .optimization import BertAdam
from .synth import load_synth, synth_image2d, synth_fundus_mask, synth_brats
from .dataloaders.datasets2d import fundus_map_mask, polyp_map_mask
from .dataloaders.datasets3d import brats_map_label

# the BASELINE.json configs (flags as in SURVEY.md 8(d)); 'bs' = per-GPU batch of the benchmark
CONFIGS = {
    'cfg1': dict(dim=2, task='fundus', num_classes=3, translayers=1, compress=[1, 1], attractors=256, bs=2, size=(256, 256)),
    'cfg2': dict(dim=2, task='fundus', num_classes=3, translayers=3, compress=[1, 1, 2, 2], attractors=256, bs=6, size=(512, 512)),
    'cfg3': dict(dim=2, task='polyp', num_classes=2, translayers=3, compress=[1, 1, 2, 2], attractors=256, bs=6, size=(352, 352)),
    'cfg4': dict(dim=3, task='brats', num_classes=4, translayers=1, compress=[1, 1], attractors=1024, bs=4, size=(112, 112, 96)),
    'cfg5': dict(dim=3, task='brats', num_classes=4, translayers=2, compress=[1, 1, 1], attractors=1024, bs=4, size=(128, 128, 128)),
}
BCE_WEIGHT = {'fundus': [0., 1., 2.], 'polyp': [0., 1.], 'brats': [0., 3., 1., 1.75]}   # train2d.py:293,344; train3d.py:223
DEFAULTS = dict(lr=2e-4, decay=1e-4, grad_clip=0.1, dropout_prob=0.2, num_modes=4)        # train2d.py:266-385 (segtran)


def model_args(c, device, dropout_prob=None, attractors=None, **over):
    a = dict(num_classes=c['num_classes'], num_attractors=attractors or c['attractors'], num_translayers=c['translayers'],
             translayer_compress_ratios=list(c['compress']), use_pretrained=False, bb_feat_upsize=True, in_fpn_use_bn=False,
             use_squeezed_transformer=True, num_modes=4, trans_output_type='private', mid_type='shared',
             pos_code_type='lsinu', pos_code_weight=1.0, pos_bias_radius=7, ablate_multihead=False, out_fpn_do_dropout=False,
             has_FFN_in_squeeze=False, attn_clip=500, qk_have_bias=True, tie_qk_scheme='shared', device=str(device),
             eval_robustness=False, use_attn_consist_loss=False, use_mince_transformer=False, mince_scales=None,
             mince_channel_props=None, dropout_prob=DEFAULTS['dropout_prob'] if dropout_prob is None else dropout_prob,
             in_fpn_layers='34', out_fpn_layers='1234', in_fpn_scheme='AN', out_fpn_scheme='AN')
    if c['dim'] == 2:
        a.update(backbone_type='eff-b4', num_modalities=0, use_global_bias=False)
    else:
        a.update(backbone_type='i3d', orig_in_channels=4, inchan_to3_scheme='bridgeconv', D_groupsize=1, D_pool_K=2,
                 out_fpn_upsampleD_scheme='interp', input_scale=(1, 1, 1))
    unknown = set(over) - set(a)
    assert not unknown, 'unknown model option(s): %s' % sorted(unknown)
    a.update(over)                      # e.g. use_squeezed_transformer=False, pos_code_type='bias' (--nosqueeze --pos bias)
    return Namespace(**a)


def build_model(cfg, device, dropout_prob=None, attractors=None, synth=True, **over):
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    args = model_args(c, device, dropout_prob, attractors, **over)
    if c['dim'] == 2:
        from .networks.segtran2d import Segtran2d, CONFIG
    else:
        from .networks.segtran3d import Segtran3d as Segtran2d, CONFIG
    CONFIG.update_config(args)
    net = Segtran2d(CONFIG)
    if synth:
        load_synth(net)
    return net.to(device)


def polyformer_optimized_params(net, poly_opt_mode, is_segtran=True, adda=False):
    """train2d.py:463-503: which parameters a --polyformer source|target run optimises.  poly_opt_mode: 'allnet' or a comma list of
    'allpoly' (every Squeeze-and-Expansion / Polyformer layer), 'inator' (the in-squeeze attention), 'k' / 'v' / 'q' (its key / value / query
    projections), 'h' (the U-Net's class projection); the domain discriminator (unless ADDA) and the reconstruction head ride along."""
    import itertools
    if poly_opt_mode == 'allnet':
        return [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    layers = net.voxel_fusion.translayers if is_segtran else net.polyformer.polyformer_layers
    pick = {'allpoly': lambda: [layers.named_parameters()],
            'inator': lambda: [t.in_ator_trans.named_parameters() for t in layers],
            'k': lambda: [t.in_ator_trans.key.named_parameters() for t in layers],
            'v': lambda: [t.in_ator_trans.out_trans.first_linear.named_parameters() for t in layers],
            'q': lambda: [t.in_ator_trans.query.named_parameters() for t in layers],
            'h': lambda: [net.outc.named_parameters()]}
    chosen = []
    for mode in poly_opt_mode.split(','):
        if mode not in pick:
            raise ValueError('unknown polyformer optimisation mode %r' % mode)
        chosen += pick[mode]()
    params = list(itertools.chain.from_iterable(chosen))
    if getattr(net, 'discriminator', None) is not None and not adda:
        params += list(net.discriminator.named_parameters())
    if getattr(net, 'recon', None) is not None:
        params += list(net.recon.named_parameters())
    return params


def init_optimizer(net, c_or_task, t_total=10000, warmup_steps=500, lr=None, decay=None, grad_clip=None, polyformer_mode=None, poly_opt_mode='allpoly',
                   bn_opt_scheme=None, adda=False):
    """Param groups as train2d.py:448-557: names containing 'backbone' get decay x 0.1, names containing 'alphas' lr x 100 without decay.
    polyformer_mode ('source' | 'target'): only the parameters polyformer_optimized_params selects are optimised and weight decay is 0
    (:463-464); bn_opt_scheme='affine' adds every BatchNorm2d's affine parameters (:505-511)."""
    lr = DEFAULTS['lr'] if lr is None else lr
    decay = DEFAULTS['decay'] if decay is None else decay
    if polyformer_mode:
        decay = 0
        named = polyformer_optimized_params(net, poly_opt_mode, hasattr(net, 'voxel_fusion'), adda)
    else:
        named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    if bn_opt_scheme == 'affine':
        have = {id(p) for _, p in named}
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                named += [(n, p) for n, p in m.named_parameters() if id(p) not in have]
    low = [p for n, p in named if 'backbone' in n]
    high = [p for n, p in named if 'backbone' not in n and 'alphas' in n]
    normal = [p for n, p in named if 'backbone' not in n and 'alphas' not in n]
    # ALWAYS the reference's four groups in its order -- normal, low_decay, (empty) no_decay, high_lr (train2d.py:536-541, train3d.py:334-339) -- so
    # that a reference checkpoint's 'optim_state' (four param_groups) loads through Optimizer.load_state_dict, which matches groups by position
    groups = [dict(params=normal, weight_decay=decay, lr=lr), dict(params=low, weight_decay=decay * 0.1, lr=lr),
              dict(params=[], weight_decay=0.0, lr=lr), dict(params=high, weight_decay=0.0, lr=lr * 100)]
    warmup_steps = min(warmup_steps, t_total // 2)
    return BertAdam(groups, lr=lr, warmup=warmup_steps / t_total, t_total=t_total, weight_decay=decay,
                    global_grad_clip=DEFAULTS['grad_clip'] if grad_clip is None else grad_clip)


def loss_weights(task, device):
    w = torch.tensor(BCE_WEIGHT[task], dtype=torch.float32)
    nc = len(BCE_WEIGHT[task])
    pos_weight = (w * (nc - 1) / w.sum()).to(device)                           # train2d.py:813-814
    cw = torch.ones(nc); cw[0] = 0; cw = (cw / cw.sum()).to(device)           # train2d.py:1123-1127
    return pos_weight, cw


def synth_batch(cfg, B, device, seed=1337):
    """(input, raw mask/label) in the encodings the reference's loaders deliver, resident on `device`."""
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    if c['dim'] == 2:
        S = c['size'][0]
        x = synth_image2d(B, S, seed)
        m = synth_fundus_mask(B, S, seed + 1)
        if c['task'] == 'polyp':
            m = m[:, :1].repeat(1, 3, 1, 1)
        return x.to(device), m.to(device)
    x, lab = synth_brats(B, *c['size'], seed=seed)
    return x.to(device), lab.to(device)


def map_mask(task, raw, exclusive=False):
    """In-step label -> n-hot map.  On the GPU this is libsegx's label kernel; the torch expressions in
    segtran_amd/dataloaders (same semantics, used by CPU-side tooling) are the readable statement of it.
    exclusive: train2d.py --exclusive (fundus only, datasets2d.py:110-111)."""
    if raw.is_cuda:
        return SF.label_nhot(raw, task, exclusive)
    if task == 'fundus':
        return fundus_map_mask(raw, exclusive)
    if task == 'polyp':
        return polyp_map_mask(raw)
    return brats_map_label(raw, False)


class TrainStep:
    """One data-parallel train step (train2d.py:1147-1337 / train3d.py:708-768 for --net segtran)."""

    def __init__(self, net, optimizer, task, reducer=None, dice_w=0.5, exclusive=False, augment=None):
        """dice_w: args.MAX_DICE_W (train2d.py:1223 / train3d.py:735, --diceweight); exclusive: --exclusive fundus masks;
        augment: optional callable (x, nhot_mask) -> (x, nhot_mask) applied on the device before the forward pass
        (train3d.py:713-715 RandomResizedCrop under --randscale)."""
        self.net, self.opt, self.task, self.reducer = net, optimizer, task, reducer
        self.dice_w, self.exclusive, self.augment = float(dice_w), bool(exclusive), augment
        if reducer is None and hasattr(optimizer, 'release_flat_grads') and getattr(optimizer, '_tabs', None) is None:
            optimizer.release_flat_grads()          # single process: no flat bucket needed, no per-parameter accumulate kernels
        dev = next(net.parameters()).device
        self.pos_weight, self.class_w = loss_weights(task, dev)
        self.stats = None
        self._bs = None

    def __call__(self, x, raw_mask):
        if self.reducer is not None:
            # EVERY step, on every rank: a ragged last batch shows up on SOME ranks only, and a check entered by those alone would pair its
            # collective with the others' gradient / BatchNorm collectives (hang or garbage instead of the intended error)
            from .dist import check_equal_batch
            check_equal_batch(x.shape[0], self.reducer.group, self.reducer.host_group)
        mask = map_mask(self.task, raw_mask, self.exclusive)
        if self.augment is not None:
            x, mask = self.augment(x, mask)
        out = self.net(x)
        if out.shape[2:] != mask.shape[2:]:
            out = SF.interp_linear(out, mask.shape[2:])          # train2d.py:1219 / train3d.py:731
        loss, self.stats = SF.seg_loss(out, mask, self.pos_weight, self.class_w, self.dice_w)
        self.opt.zero_grad()
        loss.backward()
        if self.reducer is not None:
            self.reducer.allreduce_grads()
        self.opt.step()
        return loss


def attach_adversarial(net, adversarial_mode, num_classes, num_feat_dis_in_chan=64, adda=False, recon_w=0.0, device=None, num_base_chan=32):
    """train2d.py:876-882, 923-926, 1044-1045: the domain discriminator (input: the network's last feature map, `--adv feat`, or the soft masks,
    `--adv mask`; gradient reversal in front unless ADDA trains it with its own optimizer) and the optional 1x1 reconstruction head, attached to
    the network as `net.discriminator` / `net.recon` (None when unused), so that checkpoints and polyformer_optimized_params see them."""
    from .networks.discriminator import Discriminator
    dev = device or next(net.parameters()).device
    net.discriminator = None
    if adversarial_mode:
        if adversarial_mode not in ('feat', 'mask'):
            raise ValueError("--adv must be 'feat' or 'mask' (train2d.py:88-90), got %r" % adversarial_mode)
        chan = num_classes if adversarial_mode == 'mask' else num_feat_dis_in_chan
        net.discriminator = Discriminator(chan, num_classes=1, do_revgrad=not adda, num_base_chan=num_base_chan).to(dev)
    net.recon = torch.nn.Conv2d(num_feat_dis_in_chan, 3, kernel_size=1).to(dev) if recon_w > 0 else None
    if hasattr(net, 'keep_feature_maps'):
        net.keep_feature_maps = bool(adversarial_mode) or recon_w > 0            # the loop reads net.feature_maps[-1]
    return net


def domain_adversarial_loss(net, adversarial_mode, image_batch, source_image_batch, outputs_soft, out_size, adda=False, discriminator_optim=None):
    """The unsupervised half of the few-shot domain-adaptation step, train2d.py:1259-1284, to be called right after `outputs = net(image_batch)`:
    the target batch's feature map is still in net.feature_maps[-1]; the SOURCE batch then goes through the network (overwriting it); source
    rows come first in the mixed batch and carry label 0, target rows label 1; the discriminator sees the feature maps (`feat`) or the soft masks
    resampled to the mask size (`mask`); unweighted BCE-with-logits.  ADDA (:1278-1283): the discriminator takes its own optimizer step on this
    loss first (graph retained), and the loss handed back -- the one the generator is trained on -- is the same scores against INVERTED labels.
    Returns (domain_loss, source_outputs)."""
    F = torch.nn.functional
    target_feat = net.feature_maps[-1]
    source_outputs = net(source_image_batch)
    source_feat = net.feature_maps[-1]
    mix = len(image_batch) + len(source_image_batch)
    domain_labels = torch.ones((mix, 1), device=image_batch.device)
    domain_labels[:len(source_image_batch)] = 0
    if adversarial_mode == 'feat':
        mix_dom_feat = torch.cat([source_feat, target_feat], dim=0)
    elif adversarial_mode == 'mask':
        so = source_outputs if tuple(source_outputs.shape[2:]) == tuple(out_size) else SF.interp_linear(source_outputs, tuple(out_size))
        mix_dom_feat = torch.cat([torch.sigmoid(so), outputs_soft], dim=0)
    else:
        raise ValueError(adversarial_mode)
    domain_scores = net.discriminator(mix_dom_feat)
    domain_loss = F.binary_cross_entropy_with_logits(domain_scores, domain_labels)
    if adda:
        discriminator_optim.zero_grad()
        domain_loss.backward(retain_graph=True)
        discriminator_optim.step()
        domain_loss = F.binary_cross_entropy_with_logits(domain_scores, 1 - domain_labels)
    return domain_loss, source_outputs


class AdversarialTrainStep:
    """One step of the few-shot / adversarial recipe of train2d.py (--adv feat|mask [--adda] [--reconweight]; :1147-1186 batches, :1204-1257 supervised
    part and reconstruction, :1259-1284 domain loss, :1314-1326 total loss, clip, step) as host control flow over the same kernels as TrainStep.

    step(image, raw_mask, target_unsup_image, source_image): the supervised batch (may be None with supervised_w == 0) is concatenated IN FRONT of
    the unsupervised target batch (:1166-1170); losses are computed on the first SUP_B rows only (:1228-1241); every loss term keeps the
    reference's weight: loss = SUPERVISED_W * ((1 - DICE_W) ce + DICE_W dice) + DOMAIN_LOSS_W * domain + RECON_W * recon."""

    def __init__(self, net, optimizer, task, adversarial_mode, adda=False, discriminator_optim=None, supervised_w=1.0, domain_w=0.002, recon_w=0.0,
                 dice_w=0.5, exclusive=False):
        if adversarial_mode and getattr(net, 'discriminator', None) is None:
            raise ValueError('attach_adversarial(net, ...) first: the network carries no discriminator')
        if adda and discriminator_optim is None:
            raise ValueError('--adda trains the discriminator with its own optimizer (train2d.py:1070-1073)')
        self.net, self.opt, self.task, self.mode, self.adda, self.dis_opt = net, optimizer, task, adversarial_mode, bool(adda), discriminator_optim
        self.sup_w, self.dom_w, self.recon_w, self.dice_w, self.exclusive = float(supervised_w), float(domain_w), float(recon_w), float(dice_w), bool(exclusive)
        for o in (optimizer, discriminator_optim):
            if o is not None and hasattr(o, 'release_flat_grads') and getattr(o, '_tabs', None) is None:
                o.release_flat_grads()
        dev = next(net.parameters()).device
        self.pos_weight, self.class_w = loss_weights(task, dev)
        self.stats = self.parts = None

    def __call__(self, image, raw_mask, target_unsup_image, source_image):
        F = torch.nn.functional
        if self.mode and (target_unsup_image is None or source_image is None):
            raise ValueError("--adv %s needs the unsupervised target batch and the source batch (train2d.py:1152-1165)" % self.mode)
        if not self.mode and image is None:
            raise ValueError('without --adv the step needs the supervised image batch')
        if self.sup_w > 0 and (image is None or raw_mask is None):
            raise ValueError('SUPERVISED_W > 0 needs the supervised batch and its masks (train2d.py:1232-1241)')
        sup_b = len(image) if (image is not None and self.sup_w > 0) else 0
        if self.mode:
            batch = torch.cat([image, target_unsup_image], dim=0) if sup_b > 0 else target_unsup_image       # :1166-1170
        else:
            batch = image
        out = self.net(batch)
        zero = torch.zeros((), device=batch.device)
        sup_loss, out_size = zero, tuple(out.shape[2:])
        mask = None
        if raw_mask is not None:
            # the reference maps the mask batch and resamples the outputs to ITS size whether or not the supervised loss is on (:1172-1177, 1218-1220): the
            # discriminator of `--adv mask` sees mask-sized soft masks also at SUPERVISED_W == 0
            mask = map_mask(self.task, raw_mask, self.exclusive)
            out_size = tuple(mask.shape[2:])
            if tuple(out.shape[2:]) != out_size:
                out = SF.interp_linear(out, out_size)                                  # :1219
        if sup_b > 0:
            sup_loss, sup_stats = SF.seg_loss(out[:sup_b].contiguous(), mask, self.pos_weight, self.class_w, self.dice_w)
        recon_loss = zero
        if self.recon_w > 0:                                                           # before the source pass overwrites the feature map (:1251-1257)
            recon_loss = F.mse_loss(batch, self.net.recon(self.net.feature_maps[-1]))
        domain_loss = zero
        if self.mode:
            domain_loss, _ = domain_adversarial_loss(self.net, self.mode, batch, source_image, torch.sigmoid(out) if self.mode == 'mask' else None,
                                                     out_size, self.adda, self.dis_opt)
        loss = self.sup_w * sup_loss + self.dom_w * domain_loss + self.recon_w * recon_loss       # :1314-1318
        self.parts = dict(supervised=sup_loss.detach(), domain=domain_loss.detach(), recon=recon_loss.detach())
        # what the loop logs (train_common.run): ALWAYS a tensor laid out like TrainStep.stats -- [loss, ce, dice_total, dice_c0, ...] -- with the TOTAL
        # loss in front; unsupervised-only steps (--supweight 0) carry zeros in the supervised slots (the reference logs zeros there too, :1243-1245)
        n_cls = int(self.class_w.numel())
        if sup_b > 0:
            self.stats = torch.cat([loss.detach().reshape(1), sup_stats.detach()[1:]])
        else:
            self.stats = torch.cat([loss.detach().reshape(1), torch.zeros(2 + n_cls, device=batch.device)])
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss


class GraphedTrainStep:
    """One train step captured into a hipGraph and replayed (single process): ~2300 launches (cfg2) / ~1500 (cfg1) become one graph launch,
    which removes the host from the step -- what the small per-GPU batches need (cfg1 at batch 2 is launch-bound when run eagerly).

    What makes the captured step replayable: (a) dropout -- every kernel adds a device-side base to its Philox offset (segx_set_rng_base) and
    the graph ends with segx_rng_advance(base, span of the step), so each replay draws fresh masks while forward and backward of one replay
    still regenerate the same ones; (b) the LR schedule -- folded into the optimizer's device lr table, refreshed by one small async copy
    before each replay (BertAdam.prepare_replay); (c) gradients / activations live at fixed addresses in the graph's private memory pool, so
    the optimizer's gradient pointer table is constant; (d) inputs are copied into static buffers.  Nothing in the step reads a device value
    on the host (no `.item()`), so capture needs no special casing in the model."""

    def __init__(self, step, x, raw, warmup=3):
        from . import segx
        assert step.reducer is None, 'GraphedTrainStep is the single-process step (collectives are not captured)'
        self.step, self.L = step, segx.lib()
        assert self.L.gemm_prof is None, 'per-launch event profiling cannot run inside a captured step'
        self.x, self.raw = x.clone(), raw.clone()
        self.rng_base = torch.zeros(1, dtype=torch.int64, device=x.device)
        self.L.set_rng_base(self.rng_base)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # eager warm-up: optimizer tables, the set of trained parameters (N3), allocator pools
            for _ in range(warmup):
                step(self.x, self.raw)
        torch.cuda.current_stream().wait_stream(side)
        step.opt.enter_graph_mode()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        off0 = SF._Rng.offset
        with torch.cuda.graph(self.graph):
            self.loss = step(self.x, self.raw)
            self.span = SF._Rng.offset - off0
            self.L.rng_advance(self.rng_base, self.span)
        self.replays = 0

    def __call__(self, x, raw):
        if x is not self.x:
            self.x.copy_(x, non_blocking=True)
        if raw is not self.raw:
            self.raw.copy_(raw, non_blocking=True)
        segx.lib().team_check()                     # as BertAdam.step does on the eager path (the captured optimizer step runs no host code)
        self.step.opt.prepare_replay()
        self.graph.replay()
        self.replays += 1
        return self.loss

    def check(self):
        """Call at a synchronisation point (logging, before saving a checkpoint): the captured optimizer step runs no host code, so a team exchange that timed out
        inside the LAST replay has already applied its NaN gradients -- this raises then, and the model must be reloaded from the last good checkpoint (the
        BatchNorm running statistics are not touched by a failed exchange, the weights and moments are)."""
        torch.cuda.current_stream().synchronize()
        segx.lib().team_check()

    @property
    def stats(self):
        return self.step.stats

    def close(self):
        self.L.set_rng_base(None)
