"""`python -m segtran_amd.train3d --task brats --net segtran --translayers 1 --attractors 1024 --bs 4 ...`
Mirror of the reference's code/train3d.py for the `--net segtran --segtran 3d` path (flags train3d.py:51-172; forced
backbone 'i3d' and 'bridgeconv' input bridge, train3d.py:180-195; the latent NameError on `attn_consist_loss`
(train3d.py:752-756) does not exist here: the term is simply 0)."""
import argparse
from . import train_common as tc


def main(argv=None):
    p = tc.common_flags(argparse.ArgumentParser(description=__doc__), 3)
    p.add_argument('--segtran', dest='segtran_type', type=str, default='3d')
    p.add_argument('--patch', dest='orig_patch_size', type=str, default='112,112,96')
    p.add_argument('--randscale', type=float, default=0.0,
                   help='random rescale by a factor in [1 - v, 1 + v] + pad/crop back to the patch size, on the device (RandomResizedCrop, train3d.py:713-715)')
    args = tc.finalize_args(p.parse_args(argv), 3)
    if args.segtran_type != '3d' or args.task_name != 'brats':
        raise SystemExit("only --segtran 3d --task brats is built")
    if not 0 <= args.randscale < 1:
        raise SystemExit('--randscale must lie in [0, 1)')
    size = tuple(int(v) for v in args.orig_patch_size.split(','))
    cfg = tc.make_cfg(args, 3, size, 4)
    return tc.run(args, cfg)


if __name__ == '__main__':
    main()
