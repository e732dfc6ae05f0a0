"""One convolution shape, a few launches of one kernel form (for rocprofv3 counter passes): python tools/conv_one.py <fwd|wgrad> Cin Cout D H W [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx
kind, Cin, Cout, D, H, W = sys.argv[1], *[int(v) for v in sys.argv[2:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device('cuda', 0)
L = segx.lib(); L.set_engine('x6'); L.c.segx_tune(17, 1)
B = 4
g = torch.Generator(device='cpu').manual_seed(0)
geom = (Cin, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1)
x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
if kind == 'fwd':
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1).to(dev)
    wq = L.conv3d_halo_pack(w, Cout, Cin, 0)
    y = torch.empty(B, Cout, D, H, W, device=dev)
    for _ in range(reps):
        L.conv3d_halo_fwd(x, wq, y, B, Cout, geom)
else:
    dy = torch.randn(B, Cout, D, H, W, generator=g).to(dev)
    dw = torch.empty(Cout, Cin, 3, 3, 3, device=dev)
    for _ in range(reps):
        L.conv3d_halo_wgrad(dy, x, dw, B, Cout, geom)
torch.cuda.synchronize()
