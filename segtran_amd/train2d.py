"""`python -m segtran_amd.train2d --task fundus --net segtran --bb eff-b4 --translayers 3 --layercompress 1,1,2,2 --bs 6 ...`
Mirror of the reference's code/train2d.py for the `--net segtran` path (flags train2d.py:58-243)."""
import argparse
from . import train_common as tc

NUM_CLASSES = {'fundus': 3, 'polyp': 2}            # train2d.py:286-345
DEFAULT_SIZE = {'fundus': 576, 'polyp': 320}       # orig_input_size defaults; --patch defaults to it


def main(argv=None):
    p = tc.common_flags(argparse.ArgumentParser(description=__doc__), 2)
    p.add_argument('--bb', dest='backbone_type', type=str, default='eff-b4')
    p.add_argument('--insize', dest='orig_input_size', type=str, default=None)
    p.add_argument('--patch', dest='patch_size', type=str, default=None)
    p.add_argument('--exclusive', dest='use_exclusive_masks', action='store_true')
    args = tc.finalize_args(p.parse_args(argv), 2)
    if args.use_exclusive_masks and args.task_name != 'fundus':
        raise SystemExit('--exclusive only changes the fundus label map (datasets2d.py:110-111)')
    if args.task_name not in NUM_CLASSES:
        raise SystemExit("--task %s: only 'fundus' and 'polyp' are wired (BASELINE configs)" % args.task_name)
    if args.backbone_type != 'eff-b4' and not args.backbone_type.startswith('eff-b'):
        raise SystemExit('--bb %s: only EfficientNet-V1 backbones are built' % args.backbone_type)
    S = int(str(args.patch_size or args.orig_input_size or DEFAULT_SIZE[args.task_name]).split(',')[0])
    cfg = tc.make_cfg(args, 2, (S, S), NUM_CLASSES[args.task_name])
    return tc.run(args, cfg)


if __name__ == '__main__':
    main()
