"""A/B of the pre-split B operand (segx_gemm_desc.b_planes) on the wave-specialised kernels (GPU box):  python tools/pre_bench.py
Per shape: tiles 256x128 / 128x256 with B split in registers, with B planes made once (GEMM alone), and planes made per call (split + GEMM)."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

dev = torch.device('cuda', 0)
L = segx.lib(); L.set_engine('x6')
g = torch.Generator(device='cpu').manual_seed(0)
# name, M, N, K, akc, bkc, nb, B shared over the batch, splitk
SHAPES = [('group_linear fwd NT', 24576, 1792, 1792, True, True, 4, False, 1), ('group_linear dX NN', 24576, 1792, 1792, True, False, 4, False, 1),
          ('l2 linear 896 NT', 24576, 896, 896, True, True, 4, False, 1), ('l2 linear 896 dX', 24576, 896, 896, True, False, 4, False, 1),
          ('shared linear 24576x1792x1792', 24576, 1792, 1792, True, True, 1, True, 1), ('scores QK^T', 4096, 256, 1792, True, False, 24, False, 1),
          ('scores QK^T NT', 4096, 256, 1792, True, True, 24, False, 1), ('3d outfpn 832 comp', 832, 37632, 480, True, False, 4, False, 1)]


def timed(fn, reps=6, rounds=5):
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


for name, M, N, K, akc, bkc, nb, shared, sk in SHAPES:
    A = torch.randn(nb, M, K, generator=g).to(dev) if akc else torch.randn(nb, K, M, generator=g).to(dev)
    nbB = 1 if shared else nb
    B = torch.randn(nbB, N, K, generator=g).to(dev) if bkc else torch.randn(nbB, K, N, generator=g).to(dev)
    C = torch.empty(nb, M, N, device=dev)
    a = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b = (0, 0 if shared else N * K, K, 1) if bkc else (0, 0 if shared else N * K, 1, N)
    args = (A, B, C, M, N, K, a, b, (0, M * N, N))
    fl = 2.0 * M * N * K * nb
    print('%-30s M=%6d N=%6d K=%5d nb=%2d %s%s' % (name, M, N, K, nb, 'NT'[0] if akc else 'T', 'T' if bkc else 'N'), flush=True)
    ref = None
    for tile, tn in ((segx.TILE_256x128, 'ws256x128'), (segx.TILE_WS128x256, 'ws128x256')):
        mk = lambda: L.x6_presplit(B, N, K, b[2], b[3], nb=(1, nb), s_b=(b[0], b[1]))
        planes = mk()
        t0 = timed(lambda: L.gemm(*args, nb=(1, nb), tile=tile))
        c0 = C.clone()
        t1 = timed(lambda: L.gemm(*args, nb=(1, nb), tile=tile, b_planes=planes))
        same = torch.equal(c0, C)
        t2 = timed(lambda: L.gemm(*args, nb=(1, nb), tile=tile, b_planes=mk()))
        ts = timed(mk)
        print('    %-10s split in registers %7.3f ms %6.1f TF | planes %7.3f ms %6.1f TF (%+.1f %%) | planes made per call %7.3f ms (%+.1f %%; split alone %.3f ms)  identical %s' % (
            tn, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, 100 * (t0 / t1 - 1), t2, 100 * (t0 / t2 - 1), ts, same), flush=True)
