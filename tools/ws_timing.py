"""Cycle stamps inside the wave-specialised GEMM kernels (bench-only build: python tools/build_variant.py timing gemm.hip -DSEGX_PROBE_TIMING):
python tools/ws_timing.py  -- per stage, for one consumer and one producer wave of workgroup 0: cycles waiting at the barrier, cycles issuing the
matrix instructions (consumer); barrier wait, issuing the global loads, split + LDS stores (producer).  s_memtime ticks."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx
dev = torch.device('cuda', 0)
L = segx.SegxLib(os.path.join(ROOT, 'tools', 'variants', 'libsegx_timing.so')); L.set_engine('x6')
g = torch.Generator(device='cpu').manual_seed(0)
M, N, K, nb = 24576, 1792, 1792, 4
A = torch.randn(nb, M, K, generator=g).to(dev); B = torch.randn(nb, N, K, generator=g).to(dev); C = torch.empty(nb, M, N, device=dev)
for h3 in (False, True):
    for zero in (False, True):
        if zero:
            A.zero_(); B.zero_()
        T = torch.zeros(2 * 48 * 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            T.zero_()
            L.gemm(A, B, C, M, N, K, (0, M * K, K, 1), (0, N * K, K, 1), (0, M * N, N), nb=(1, nb), tile=6, f16x3=h3, aux=T.view(torch.float32))
        torch.cuda.synchronize()
        t = T.cpu().view(2, 48, 8)
        c = t[0, 4:44]; p = t[1, 2:14]
        cw = (c[:, 1] - c[:, 0]).float(); cm = (c[:, 2] - c[:, 1]).float(); cs = (c[1:, 0] - c[:-1, 0]).float()
        line = '%-7s %-6s consumer: stage %.0f ticks = barrier wait %.0f + fragment reads / matrix issue %.0f' % ('f16x3' if h3 else 'bf16x6', 'zeros' if zero else 'randn',
                                                                                                                cs.median(), cw.median(), cm.median())
        if h3:
            pw = (p[:, 1] - p[:, 0]).float(); pl = pw * 0; ps = (p[:, 3] - p[:, 1]).float(); pp = (p[1:, 0] - p[:-1, 0]).float() / 3
            line += ' | producer: stage %.0f = barrier wait %.0f + (%.0f) + interleaved loads and split / store %.0f' % (pp.median(), pw.median(), pl.median(), ps.median())
        print(line, flush=True)
    A = torch.randn(nb, M, K, generator=g).to(dev); B = torch.randn(nb, N, K, generator=g).to(dev)
