"""3 x 3 x 3 stride-1 'same' convolutions of Inception-I3D (cfg4: 112 x 112 x 96, cfg5: 128^3, batch 4): the LDS-resident-halo kernel (conv3d_halo.hip, every
output-channel tile it has) against the im2col kernel of conv3d.hip, forward and data-gradient shapes.  GPU box:  python tools/conv_bench.py [cfg4|cfg5|all]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

dev = torch.device('cuda', 0)
L = segx.lib()
L.set_engine('x6')
# (name, Cin, Cout, (D, H, W)): the b1b / b2b convolutions (aj_i3d.py:198-273) and Conv3d_2c; the data gradient of (Cin, Cout) is the forward shape (Cout, Cin)
LAYERS = [('2c', 64, 192), ('3b.b1b', 96, 128), ('3b.b2b', 16, 32), ('3c.b1b', 128, 192), ('3c.b2b', 32, 96), ('4b.b1b', 96, 208), ('4b.b2b', 16, 48),
          ('4c.b1b', 112, 224), ('4c.b2b', 24, 64), ('4d.b1b', 128, 256), ('4e.b1b', 144, 288), ('4e.b2b', 32, 64), ('4f.b1b', 160, 320), ('4f.b2b', 32, 128),
          ('5b.b1b', 160, 320), ('5c.b1b', 192, 384), ('5c.b2b', 48, 128)]
STAGE = {'2': 0, '3': 1, '4': 2, '5': 3}
SIZES = {'cfg4': [(48, 56, 56), (48, 28, 28), (24, 14, 14), (12, 7, 7)], 'cfg5': [(64, 64, 64), (64, 32, 32), (32, 16, 16), (16, 8, 8)]}


def timed(fn, reps=5, rounds=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    B = 4
    g = torch.Generator(device='cpu').manual_seed(0)
    L.c.segx_tune(17, 1)
    for cfg in (['cfg4', 'cfg5'] if which == 'all' else [which]):
        for name, ci, co in LAYERS:
            D, H, W = SIZES[cfg][STAGE[name[0]]]
            for tag, Cin, Cout in (('fwd', ci, co), ('dgrad', co, ci)):
                geom = (Cin, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
                x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
                w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1).to(dev)
                flops = 2.0 * B * Cout * D * H * W * Cin * 27
                y0 = torch.empty(B, Cout, D, H, W, device=dev)
                wp = torch.empty_like(w)
                L.conv3d_pack_weights(w, wp, Cout, Cin, 27, 0)
                sk = L.conv3d_splitk(B, Cout, geom, False)
                ws = torch.empty(sk * y0.numel(), device=dev) if sk > 1 else None
                t0 = timed(lambda: L.conv3d_fwd(x, wp, y0, B, Cout, geom, sk, ws, packed=True))
                line = '%s %-7s %-5s Cin %3d Cout %3d %2dx%2dx%2d | im2col(sk %2d) %7.3f ms %6.1f TF' % (cfg, name, tag, Cin, Cout, D, H, W, sk, t0, flops / t0 / 1e9)
                if L.conv3d_halo_ok(B, Cout, geom):
                    wq = L.conv3d_halo_pack(w, Cout, Cin, 0)
                    tp = timed(lambda: L.conv3d_halo_pack(w, Cout, Cin, 0))
                    for mt in (64, 128, 192):
                        if mt > 64 and mt - 64 >= Cout + 63:
                            continue
                        y1 = torch.empty_like(y0)
                        t1 = timed(lambda: L.conv3d_halo_fwd(x, wq, y1, B, Cout, geom, mtile=mt))
                        err = (y1 - y0).abs().max().item() / max(y0.abs().max().item(), 1e-20)
                        line += ' | halo m%3d %7.3f ms %6.1f TF%s' % (mt, t1, flops / t1 / 1e9, '' if err < 2e-5 else ' MISMATCH %.1e' % err)
                    line += ' | pack %.3f ms' % tp
                else:
                    line += ' | halo: not served'
                print(line, flush=True)
    L.c.segx_tune(17, 256)


def wgrad():
    """weight gradients: resident-halo form against the im2col kernels (+ their batch sum and un-packing launches)"""
    which = sys.argv[2] if len(sys.argv) > 2 else 'all'
    B = 4
    g = torch.Generator(device='cpu').manual_seed(0)
    L.c.segx_tune(17, 1)
    for cfg in (['cfg4', 'cfg5'] if which == 'all' else [which]):
        for name, Cin, Cout in LAYERS:
            D, H, W = SIZES[cfg][STAGE[name[0]]]
            geom = (Cin, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
            x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
            dy = torch.randn(B, Cout, D, H, W, generator=g).to(dev)
            N = Cin * 27
            flops = 2.0 * B * Cout * D * H * W * N
            sk = L.conv3d_splitk(B, Cout, geom, True)
            ws = torch.empty(sk * B * Cout * N, device=dev) if sk > 1 else None
            dwb = torch.empty(B, Cout * N, device=dev); dwp = torch.empty(Cout * N, device=dev); dw0 = torch.empty(Cout * N, device=dev)
            cw = torch.empty(L.colreduce_ws(B, Cout * N, 1), device=dev)

            def old():
                L.conv3d_bwd_weight(dy, x, dwb, B, Cout, geom, sk, ws, packed=True)
                L.colsum(dwb, dwp, cw, B, Cout * N)
                L.conv3d_unpack_wgrad(dwp, dw0, Cout, Cin, 27)
            t0 = timed(old)
            line = '%s %-7s wgrad Cin %3d Cout %3d %2dx%2dx%2d | im2col(sk %2d) %7.3f ms %6.1f TF' % (cfg, name, Cin, Cout, D, H, W, sk, t0, flops / t0 / 1e9)
            if L.conv3d_halo_wgrad_ok(B, Cout, geom):
                dw1 = torch.empty(Cout, Cin, 3, 3, 3, device=dev)
                t1 = timed(lambda: L.conv3d_halo_wgrad(dy, x, dw1, B, Cout, geom))
                err = (dw1.reshape(-1) - dw0).abs().max().item() / max(dw0.abs().max().item(), 1e-20)
                line += ' | halo %7.3f ms %6.1f TF%s' % (t1, flops / t1 / 1e9, '' if err < 2e-5 else ' MISMATCH %.1e' % err)
            else:
                line += ' | halo: not served'
            print(line, flush=True)
    L.c.segx_tune(17, 256)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
        wgrad()
        sys.exit(0)
    main()
