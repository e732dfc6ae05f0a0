"""Bandwidth of the trilinear resampling kernels at the cfg-4 out-FPN shapes.  python tools/interp_bench.py  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segtran_amd import segx
L = segx.lib(); dev = torch.device('cuda', 0)
for (planes, d, h, w, D, H, W, with_base) in [(4096, 24, 56, 56, 48, 56, 56, False), (4096, 12, 14, 14, 24, 56, 56, True),
                                              (3328, 24, 14, 14, 24, 28, 28, True), (16, 48, 56, 56, 96, 112, 112, False),
                                              (6 * 448, 1, 64, 64, 1, 256, 256, True)]:
    x = torch.randn(planes, d, h, w, device=dev); out = torch.empty(planes, D, H, W, device=dev)
    base = torch.randn(planes, D, H, W, device=dev) if with_base else None
    byt = 4.0 * (x.numel() + out.numel() * (2 if with_base else 1))
    for var in (1, 2):
        L.c.segx_tune(1, var)
        for _ in range(2): L.interp_fwd(x, base, out, planes, d, h, w, D, H, W)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): L.interp_fwd(x, base, out, planes, d, h, w, D, H, W)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print('planes %5d %s->%s base=%d variant %d: %7.3f ms  %6.0f GB/s (%.2f GB)' % (planes, (d, h, w), (D, H, W), with_base, var, ms, byt / ms / 1e6, byt / 1e9))
L.c.segx_tune(1, 0)
