"""Shared trainer skeleton for the `--net segtran` path of the reference's train2d.py / train3d.py.

Same flag names / dests / defaults for everything the segtran path reads (train2d.py:58-243, train3d.py:51-172),
same defaults table (train2d.py:266-385: BertAdam, lr 2e-4, decay 1e-4, grad_clip 0.1, dropout 0.2, 4 modes), same
launcher contract (`python -m torch.distributed.run --nproc_per_node=N -m segtran_amd.train2d ...`, `--local_rank`
/ env://, `--bs` = GLOBAL batch, `args.batch_size //= world_size`, train2d.py:791), same checkpoint wire format
(`{'iter_num', 'model', 'args'}` every `--saveiter`, rank 0; train2d.py:640-648).

Out of scope here: the file loaders / imgaug pipelines.  Batches come from
`segtran_amd.engine.synth_batch` (the BASELINE.md synthetic inputs) unless the caller passes its own iterator of
(image, raw_mask) tensors to `run()`.  Flags that select reference features outside the hot path are accepted and
rejected with a clear message instead of being silently ignored.
"""
import argparse
import logging
import os
import time
from datetime import datetime

import torch

from . import engine, dist as sdist, functional as SF

UNSUPPORTED = {'polyformer_mode': None, 'use_global_bias': False, 'ablate_multihead': False,
               'use_attn_consist_loss': False}


def common_flags(p, dim):
    p.add_argument('--task', dest='task_name', type=str, default='fundus' if dim == 2 else 'brats')
    p.add_argument('--ds', dest='ds_names', type=str, default=None, help='(file datasets are out of scope: synthetic batches are used)')
    p.add_argument('--split', dest='ds_split', type=str, default='all')
    p.add_argument('--maxiter', type=int, default=10000)
    p.add_argument('--saveiter', type=int, default=500)
    p.add_argument('--cp', dest='checkpoint_path', type=str, default=None)
    p.add_argument('--lrwarmup', dest='lr_warmup_steps', type=int, default=500)
    p.add_argument('--bs', dest='batch_size', type=int, default=6 if dim == 2 else 4, help='Total batch_size on all GPUs')
    p.add_argument('--opt', type=str, default=None)
    p.add_argument('--lr', type=float, default=-1)
    p.add_argument('--decay', type=float, default=-1)
    p.add_argument('--gradclip', dest='grad_clip', type=float, default=-1)
    p.add_argument('--attnclip', dest='attn_clip', type=int, default=500)
    p.add_argument('--local_rank', default=0, type=int)
    p.add_argument('--diceweight', dest='MAX_DICE_W', type=float, default=0.5)
    p.add_argument('--seed', type=int, default=1337)
    p.add_argument('--debug', dest='debug', action='store_true')
    p.add_argument('--net', type=str, default='segtran')
    p.add_argument('--nopretrain', dest='use_pretrained', action='store_false')
    p.add_argument('--nosqueeze', dest='use_squeezed_transformer', action='store_false')
    p.add_argument('--attractors', dest='num_attractors', default=256 if dim == 2 else 1024, type=int)
    p.add_argument('--noqkbias', dest='qk_have_bias', action='store_false')
    p.add_argument('--translayers', dest='num_translayers', default=1, type=int)
    p.add_argument('--layercompress', dest='translayer_compress_ratios', type=str, default=None)
    p.add_argument('--modes', type=int, dest='num_modes', default=-1)
    p.add_argument('--multihead', dest='ablate_multihead', action='store_true')
    p.add_argument('--dropout', type=float, dest='dropout_prob', default=-1)
    p.add_argument('--pos', dest='pos_code_type', type=str, default='lsinu')
    p.add_argument('--posw', dest='pos_code_weight', type=float, default=1.0)
    p.add_argument('--posr', dest='pos_bias_radius', type=int, default=7)
    p.add_argument('--squeezeuseffn', dest='has_FFN_in_squeeze', action='store_true')
    p.add_argument('--attnconsist', dest='use_attn_consist_loss', action='store_true')
    p.add_argument('--mince', dest='use_mince_transformer', action='store_true')
    p.add_argument('--mincescales', dest='mince_scales', type=str, default=None, help='comma list of the mince scales, e.g. 4,2,1')
    p.add_argument('--minceprops', dest='mince_channel_props', type=str, default=None, help='comma list of value-channel proportions')
    p.add_argument('--infpn', dest='in_fpn_layers', default='34')
    p.add_argument('--outfpn', dest='out_fpn_layers', default='1234')
    p.add_argument('--outdrop', dest='out_fpn_do_dropout', action='store_true')
    p.add_argument('--inbn', dest='in_fpn_use_bn', action='store_true')
    p.add_argument('--nofeatup', dest='bb_feat_upsize', action='store_false')
    if dim == 2:          # the few-shot / adversarial recipe (train2d.py:87-107); file datasets are out of scope: the source / target batches are synthetic
        p.add_argument('--adv', dest='adversarial_mode', type=str, default=None, choices=[None, 'none', 'feat', 'mask'])
        p.add_argument('--featdisinchan', dest='num_feat_dis_in_chan', type=int, default=64)
        p.add_argument('--sourcebs', dest='source_batch_size', type=int, default=-1)
        p.add_argument('--targetbs', dest='target_unsup_batch_size', type=int, default=-1)
        p.add_argument('--domweight', dest='DOMAIN_LOSS_W', type=float, default=0.002)
        p.add_argument('--supweight', dest='SUPERVISED_W', type=float, default=1)
        p.add_argument('--reconweight', dest='RECON_W', type=float, default=0)
        p.add_argument('--adda', dest='adda', action='store_true')
    p.add_argument('--tunebn', dest='tune_bn_only', action='store_true', help='only refresh the BatchNorm statistics of the first backbone stages')
    p.add_argument('--logiter', type=int, default=50, help='host-side logging period (each log line synchronises the device)')
    p.add_argument('--synthweights', dest='synth_weights', action='store_true',
                   help='(not a reference flag) start from the name-hashed synthetic weights of segtran_amd/synth.py -- what bench.py and the '
                        'parity fixtures use -- instead of the pretrained backbone (default) or the random initialisation (--nopretrain)')
    return p


def finalize_args(args, dim):
    if args.net != 'segtran':
        raise SystemExit("only --net segtran is built (the reference's other models are out of scope, SURVEY.md section 2)")
    for k, v in UNSUPPORTED.items():
        if getattr(args, k, v) != v:
            raise SystemExit('--%s selects a reference feature outside the MI355X hot path (SURVEY.md 8(f))' % k)
    if getattr(args, 'adversarial_mode', None) == 'none':                  # train2d.py:263-264
        args.adversarial_mode = None
    if args.opt not in (None, 'adamw'):
        raise SystemExit("segtran default optimiser is BertAdam ('adamw' in the reference's table); other --opt values are not built")
    d = engine.DEFAULTS
    args.lr = d['lr'] if args.lr < 0 else args.lr
    args.decay = d['decay'] if args.decay < 0 else args.decay
    args.grad_clip = d['grad_clip'] if args.grad_clip < 0 else args.grad_clip
    args.dropout_prob = d['dropout_prob'] if args.dropout_prob < 0 else args.dropout_prob
    args.num_modes = d['num_modes'] if args.num_modes < 0 else args.num_modes
    if args.translayer_compress_ratios is not None:
        args.translayer_compress_ratios = [int(c) for c in str(args.translayer_compress_ratios).split(',')]
    else:
        args.translayer_compress_ratios = [1] * (args.num_translayers + 1)
    if args.mince_scales is not None:                                      # train2d.py:254-257
        args.mince_scales = [int(v) for v in str(args.mince_scales).split(',')]
    if args.mince_channel_props is not None:
        args.mince_channel_props = [float(v) for v in str(args.mince_channel_props).split(',')]
    if getattr(args, 'tune_bn_only', False):                               # train2d.py:747-751
        if args.checkpoint_path is None:
            raise SystemExit('Tuning BN requires to specify a checkpoint to load')
        args.lr_warmup_steps = 0
    if args.use_mince_transformer and not (args.mince_scales and args.mince_channel_props
                                           and len(args.mince_scales) == len(args.mince_channel_props)):
        raise SystemExit('--mince needs --mincescales and --minceprops of equal length')
    return args


ARCH_FLAGS = ('use_squeezed_transformer', 'has_FFN_in_squeeze', 'in_fpn_use_bn', 'out_fpn_do_dropout', 'qk_have_bias', 'num_modes', 'pos_code_type', 'pos_code_weight', 'pos_bias_radius', 'attn_clip',
              'use_mince_transformer', 'mince_scales', 'mince_channel_props', 'in_fpn_layers', 'out_fpn_layers', 'bb_feat_upsize')


def arch_overrides(args):
    """The architecture flags of the command line that Segtran2d/3d's config reads (train2d.py:256-303 -> CONFIG.update_config)."""
    return {k: getattr(args, k) for k in ARCH_FLAGS if hasattr(args, k)}


def make_cfg(args, dim, size, num_classes):
    return dict(dim=dim, task=args.task_name, num_classes=num_classes, translayers=args.num_translayers,
                compress=args.translayer_compress_ratios, attractors=args.num_attractors, bs=args.batch_size, size=tuple(size))


def save_model(net, args, ckpt_dir, iter_num):
    """train2d.py:640-648 wire format."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, 'iter_%d.pth' % iter_num)
    a = {k: v for k, v in vars(args).items() if not isinstance(v, torch.Tensor)}
    torch.save({'iter_num': iter_num, 'model': net.state_dict(), 'args': a}, path)
    logging.info('save model to %s', path)
    return path


# checkpoint arguments that may differ from the command line (train2d.py:584-600 / train3d.py:372-387, merged)
IGNORED_CP_ARGS = {'maxiter', 'checkpoint_path', 'model_input_size', 't_total', 'num_workers', 'lr_warmup_ratio', 'lr_warmup_steps', 'local_rank',
                   'distributed', 'world_size', 'saveiter', 'dice_warmup_steps', 'opt', 'lr', 'decay', 'initializer_range',
                   'base_initializer_range', 'grad_clip', 'localization_prob', 'tune_bn_only', 'MAX_DICE_W', 'deterministic', 'lr_schedule',
                   'out_fpn_do_dropout', 'randscale', 'do_affine', 'focus_class', 'bce_weight', 'translayer_compress_ratios', 'seed', 'debug',
                   'ds_name', 'batch_size', 'dropout_prob', 'patch_size', 'orig_input_size', 'output_upscale', 'use_pretrained', 'checkpoint_dir',
                   'iters', 'out_origsize', 'out_softscores', 'verbose_output', 'gpu', 'test_interp', 'do_remove_frag', 'reload_mask', 'ds_split',
                   'ds_names', 'train_ds_names', 'job_name', 'mean', 'std', 'mask_thres', 'sample_num', 'perturb_pew_range', 'pos_embed_every_layer',
                   'pos_in_attn_only', 'attention_mode_dim', 'ablate_pos_embed_type', 'use_attn_consist_loss', 'ATTNCONSIST_W',
                   'logiter', 'synth_weights'}            # the last two are flags of this mirror only
WARN_CP_ARGS = {'num_recurrences'}


def check_checkpoint_args(args, cp_args):
    """train2d.py:602-609 / train3d.py:389-392: every architecture-relevant argument stored in the checkpoint must equal the command line's;
    the reference prints the first inconsistent pair and exits -- here ALL of them are listed.  Keys the command line does not define
    (a reference checkpoint carries its whole argparse namespace) are skipped: the reference would die on them with a KeyError."""
    if cp_args is None:
        return
    have, bad = vars(args), []
    for k, v in cp_args.items():
        if k in IGNORED_CP_ARGS or k not in have:
            continue
        cur = have[k]
        same = (list(cur) == list(v)) if isinstance(cur, (list, tuple)) and isinstance(v, (list, tuple)) else (cur == v)
        if not same:
            if k in WARN_CP_ARGS:
                logging.warning('args[%s]=%s, checkpoint args[%s]=%s, inconsistent!', k, cur, k, v)
            else:
                bad.append('args[%s]=%s, checkpoint args[%s]=%s, inconsistent!' % (k, cur, k, v))
    if bad:
        raise SystemExit('\n'.join(bad))


def load_model(net, args, path, optimizer=None, load_optim_state=False):
    """train2d.py:567-638 / train3d.py:353-420: `{'iter_num', 'model', 'args'[, 'optim_state']}` or a bare state_dict; the checkpoint's
    architecture arguments must agree with the command line (check_checkpoint_args); 'attn_scaler' entries are dropped (:611-623); the
    checkpoint may miss keys but may not carry unknown ones or other shapes (`net.load_state_dict` raises, as in the reference).
    Returns the checkpoint's iteration count."""
    ck = torch.load(path, map_location='cpu')
    if isinstance(ck, dict) and 'model' in ck:
        sd, cp_args, cp_iter, optim_state = ck['model'], ck.get('args'), int(ck.get('iter_num', 0)), ck.get('optim_state')
    else:
        sd, cp_args, cp_iter, optim_state = ck, None, 0, None
    if isinstance(cp_args, argparse.Namespace):
        cp_args = vars(cp_args)
    if getattr(args, 'net', 'segtran') == 'segtran':
        check_checkpoint_args(args, cp_args)
    own = net.state_dict()
    dropped = [k for k in sd if 'attn_scaler' in k]
    keep = {k: v for k, v in sd.items() if 'attn_scaler' not in k and '.pos_coder.all_' not in k}   # all_*: the reference's index buffers (not built here)
    unknown = sorted(k for k in keep if k not in own)
    wrong = sorted('%s: checkpoint %s vs model %s' % (k, tuple(v.shape), tuple(own[k].shape)) for k, v in keep.items()
                   if k in own and tuple(v.shape) != tuple(own[k].shape))
    if unknown or wrong:
        raise RuntimeError('checkpoint %s does not fit the model built from the command line:\n  unexpected keys: %s\n  shape mismatches: %s'
                           % (path, unknown[:8], wrong[:8]))
    own.update(keep)
    net.load_state_dict(own)
    if load_optim_state and optimizer is not None and optim_state is not None and not dropped:      # :629-635
        optimizer.load_state_dict(optim_state)
        args.lr_warmup_steps = 0                  # the reference's bookkeeping (train2d.py:633-635); the optimizer built above keeps ITS warm-up: what
        logging.info('optimizer state restored: schedule position %d (args.lr_warmup_steps set to 0 as in the reference)',     # continues is the loaded position
                     getattr(optimizer, 'step_count', 0))
    logging.info("Model loaded from '%s' (%d/%d tensors)", path, len(keep), len(own))
    return cp_iter


def run(args, cfg, batches=None):
    """The hot loop (train2d.py:1134-1389 / train3d.py:699-811) for --net segtran."""
    rank, local, world = sdist.init_distributed()
    args.world_size, args.distributed = world, world > 1
    if not torch.cuda.is_available():
        raise SystemExit('segtran_amd trainers need an MI355X: the product path has no CPU fallback')
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    args.batch_size //= world                                             # --bs is the GLOBAL batch (train2d.py:791)
    if args.batch_size < 1:
        raise SystemExit('--bs %d is smaller than the world size %d' % (args.batch_size * world, world))
    torch.manual_seed(args.seed); SF.manual_seed(args.seed + rank)
    is_master = rank == 0
    ts = datetime.now().strftime('%m%d%H%M')
    ckpt_dir = os.path.join('..', 'model', '%s-%s-%s' % (args.net, args.task_name, ts))
    logging.basicConfig(level=logging.INFO if is_master else logging.WARNING, format='[%(asctime)s] %(message)s', datefmt='%H:%M:%S')

    # Initial weights.  Reference: the pretrained backbone (default) or, with --nopretrain, the model's own random initialisation
    # (SegtranInitWeights); a --cp checkpoint then overwrites everything.  The published backbone files cannot be fetched here (no
    # network): without them the default fails loudly instead of training something else.  --synthweights (this mirror only) selects
    # the name-hashed synthetic weights that bench.py and the parity fixtures use.
    synth = bool(getattr(args, 'synth_weights', False)) and not args.checkpoint_path
    pretrained = bool(args.use_pretrained) and not synth
    build = lambda pre: engine.build_model(cfg, dev, dropout_prob=args.dropout_prob, attractors=args.num_attractors, synth=synth,      # noqa: E731
                                           use_pretrained=pre, **arch_overrides(args))
    try:
        net = build(pretrained)
    except RuntimeError as e:
        if 'pretrained' not in str(e):
            raise
        if not args.checkpoint_path:
            raise SystemExit('%s\n(use --nopretrain for a random initialisation, --synthweights for the synthetic benchmark weights, or --cp)' % e)
        # The reference builds the pretrained network first and overlays the checkpoint (train2d.py:1028-1077), so tensors a partial checkpoint
        # misses stay pretrained.  Without the published files (no network here) a --cp run starts from the model's own initialisation instead:
        # say so -- load_model reports how many tensors the checkpoint covered.
        logging.warning('%s -- building without the pretrained backbone; tensors missing from --cp keep their random initialisation', e)
        net = build(False)
    if synth:
        logging.warning('--synthweights: training starts from SYNTHETIC name-hashed weights (segtran_amd/synth.py), not from a pretrained backbone')
    # The optimizer exists BEFORE the checkpoint is read, as in the reference (train2d.py:1066-1077, train3d.py:652-660): a checkpoint that carries
    # 'optim_state' restores the moments and the schedule position.  (The reference's own save_model comments 'optim_state' out -- train2d.py:644,
    # train3d.py:412 -- so its checkpoints restart the moments; a 3-D resume then continues the iteration count, a 2-D one restarts it.)
    adv = getattr(args, 'adversarial_mode', None) if dim_of(cfg) == 2 else None
    if adv or getattr(args, 'RECON_W', 0) > 0:                            # train2d.py:876-882, 923-926, 1044-1045: discriminator / reconstruction head ride on the network
        if world > 1:
            raise SystemExit('--adv / --reconweight: the adversarial recipe is mirrored for one process (its two optimizers are not bucketed)')
        engine.attach_adversarial(net, adv, cfg['num_classes'], args.num_feat_dis_in_chan, getattr(args, 'adda', False), args.RECON_W, dev)
    # --tunebn never takes an optimizer step: no flat moment / gradient buffers (3x the parameter memory) and no gradient hooks for it.
    tune_only = bool(getattr(args, 'tune_bn_only', False))
    opt = None if tune_only else engine.init_optimizer(net, args.task_name, t_total=args.maxiter, warmup_steps=args.lr_warmup_steps, lr=args.lr,
                                                       decay=args.decay, grad_clip=args.grad_clip)
    load_optim = dim_of(cfg) == 3 or (getattr(args, 'polyformer_mode', None) is None and getattr(args, 'opt_filters', None) is None
                                      and getattr(args, 'adversarial_mode', None) is None)           # train2d.py:1076; train3d.py:397 always
    iter_num = load_model(net, args, args.checkpoint_path, optimizer=opt, load_optim_state=load_optim) if args.checkpoint_path else 0
    if dim_of(cfg) == 2:
        iter_num = 0                                                      # train2d.py:1074-1081 (`continue_iter = False`): 2-D always restarts the count
    # 3-D resumes from the checkpoint's iteration (train3d.py:658-661)
    sdist.enable_sync_batchnorm()
    if tune_only:
        if batches is None:
            fixed = engine.synth_batch(cfg, args.batch_size, dev, seed=args.seed + rank)
            batches = iter(lambda: fixed, None)
        return tune_bn(net, args, batches, dev, ckpt_dir, is_master)
    net.train()
    if iter_num > 0 and opt.step_count == 0:
        opt.step_count = iter_num                 # no optimizer state in the checkpoint: keep the LR schedule at the resumed iteration (moments restart)
    reducer = sdist.GradReducer(opt) if world > 1 else None
    augment = None
    if dim_of(cfg) == 3 and getattr(args, 'randscale', 0):              # train3d.py:524-527, 713-715
        from .dataloaders.datasets3d import RandomResizedCrop
        crop_percents, out_size = (-args.randscale, args.randscale), tuple(cfg['size'])
        augment = lambda x, m: RandomResizedCrop(x, m, out_size, crop_percents)     # noqa: E731
    if adv or getattr(args, 'RECON_W', 0) > 0:
        from .optimization import BertAdam
        dis_opt = None
        if adv and args.adda:                                             # train2d.py:1070-1073
            dis_opt = BertAdam(net.discriminator.parameters(), lr=args.lr, warmup=min(args.lr_warmup_steps, args.maxiter // 2) / args.maxiter,
                               t_total=args.maxiter, weight_decay=args.decay)
        step = engine.AdversarialTrainStep(net, opt, args.task_name, adv, adda=getattr(args, 'adda', False), discriminator_optim=dis_opt,
                                           supervised_w=args.SUPERVISED_W, domain_w=args.DOMAIN_LOSS_W, recon_w=args.RECON_W, dice_w=args.MAX_DICE_W,
                                           exclusive=getattr(args, 'use_exclusive_masks', False))
        if batches is None:
            sb = args.source_batch_size if args.source_batch_size > 0 else args.batch_size
            tb = args.target_unsup_batch_size if args.target_unsup_batch_size > 0 else args.batch_size
            fixed = engine.synth_batch(cfg, args.batch_size, dev, seed=args.seed + rank) + \
                (engine.synth_batch(cfg, tb, dev, seed=args.seed + 101)[0], engine.synth_batch(cfg, sb, dev, seed=args.seed + 202)[0])
            batches = iter(lambda: fixed, None)                           # (image, mask, unsupervised target images, source images), train2d.py:1147-1170
    else:
        step = engine.TrainStep(net, opt, args.task_name, reducer, dice_w=args.MAX_DICE_W,
                                exclusive=getattr(args, 'use_exclusive_masks', False), augment=augment)
        if batches is None:
            fixed = engine.synth_batch(cfg, args.batch_size, dev, seed=args.seed + rank)
            batches = iter(lambda: fixed, None)
    t0 = time.time()
    try:
        return _train_loop(net, args, step, opt, batches, dev, iter_num, is_master, ckpt_dir, t0)
    finally:
        if reducer is not None:
            reducer.close()                       # hooks off, BatchNorm's process-wide form back to what it was


def _train_loop(net, args, step, opt, batches, dev, iter_num, is_master, ckpt_dir, t0):
    for batch in batches:
        iter_num += 1
        step(*(t.to(dev, non_blocking=True) for t in batch))
        if iter_num % args.logiter == 0 or iter_num == args.maxiter:
            stats = sdist.reduce_scalars(step.stats.detach())            # ONE collective for all logged scalars (C3)
            if is_master:
                s = stats.tolist()
                logging.info('%d loss: %.4f, ce: %.4f, dice: %.4f (%s), lr %.2e, %.1f it/s', iter_num, s[0], s[1], s[2],
                             ', '.join('%.3f' % v for v in s[4:]), opt.get_lr()[0], args.logiter / max(time.time() - t0, 1e-9))
            t0 = time.time()
        if is_master and (iter_num % args.saveiter == 0 or iter_num == args.maxiter):
            save_model(net, args, ckpt_dir, iter_num)
        if iter_num >= args.maxiter:
            break
    return net


def set_tune_bn_mode(net):
    """train2d.py:1089-1098: everything in eval mode except the EfficientNet blocks before endpoint 3, whose BatchNorm layers
    keep updating their running statistics.  Returns the number of blocks left in training mode."""
    net.eval()
    if not hasattr(net.backbone, '_blocks'):
        raise SystemExit("Backbone '%s' not supported by --tunebn." % getattr(net, 'backbone_type', '?'))
    stop = net.backbone.endpoint_blk_indices[3]
    for idx, block in enumerate(net.backbone._blocks):
        if idx == stop:
            break
        block.train()
    return stop


def tune_bn(net, args, batches, dev, ckpt_dir, is_master):
    """--tunebn loop (train2d.py:1195-1204): gradient-free forward passes that refresh the BatchNorm running statistics of the
    first backbone stages; a checkpoint every 50 iterations."""
    logging.info('Tuning stops at block %d', set_tune_bn_mode(net))
    iter_num = 0
    for x, _ in batches:
        iter_num += 1
        with torch.no_grad():
            net(x.to(dev, non_blocking=True))
        if is_master and (iter_num % 50 == 0 or iter_num == args.maxiter):
            save_model(net, args, ckpt_dir, iter_num)
        if iter_num >= args.maxiter:
            break
    return net


def dim_of(cfg):
    return cfg['dim']
