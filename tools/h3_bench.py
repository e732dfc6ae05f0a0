"""f16x3 (two fp16 planes, three matrix instructions per block product, gemm_h3.h) against bf16x6 on the wave-specialised kernels (GPU box):
python tools/h3_bench.py   -- time incl. the row-scale pre-pass launches, and the error of both against fp64 on sampled rows."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

dev = torch.device('cuda', 0)
L = segx.lib(); L.set_engine('x6')
g = torch.Generator(device='cpu').manual_seed(0)
SHAPES = [('group_linear fwd NT', 24576, 1792, 1792, True, True, 4, 1), ('group_linear dX NN', 24576, 1792, 1792, True, False, 4, 1),
          ('group_linear dW TN sk3', 1792, 1792, 24576, False, False, 4, 3), ('scores QK^T', 4096, 256, 1792, True, False, 24, 1),
          ('dP 1792x256x4096', 1792, 256, 4096, False, False, 24, 3), ('l2 linear 896 NT', 24576, 896, 896, True, True, 4, 1),
          ('3d translayer 16384x1024x1024', 16384, 1024, 1024, True, True, 4, 1), ('3d outfpn 480x262144x832', 480, 262144, 832, False, False, 4, 1),
          ('square 8192', 8192, 8192, 8192, True, True, 1, 1)]


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


for name, M, N, K, akc, bkc, nb, sk in SHAPES:
    if 4.0 * nb * (M * K + N * K + (1 + sk) * M * N) > 40e9:
        continue
    A = torch.randn(nb, M, K, generator=g).to(dev) if akc else torch.randn(nb, K, M, generator=g).to(dev)
    B = torch.randn(nb, N, K, generator=g).to(dev) if bkc else torch.randn(nb, K, N, generator=g).to(dev)
    C = torch.empty(nb, M, N, device=dev)
    a = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    ws = torch.empty(sk * nb * M * N, device=dev) if sk > 1 else None
    args = (A, B, C, M, N, K, a, b, (0, M * N, N))
    fl = 2.0 * M * N * K * nb
    # fp64 reference on 64 sampled rows of batch member 0
    rows = torch.randint(0, M, (64,), generator=g).to(dev)
    A0 = (A[0] if akc else A[0].t())[rows].double(); B0 = (B[0] if bkc else B[0].t()).double()
    ref = A0 @ B0.t(); mag = A0.abs() @ B0.abs().t()
    print('%-32s M=%6d N=%6d K=%5d nb=%2d sk=%d %s%s' % (name, M, N, K, nb, sk, 'NT'[0] if akc else 'T', 'T' if bkc else 'N'), flush=True)
    for tile, tn in ((segx.TILE_256x128, 'ws256x128'), (segx.TILE_WS128x256, 'ws128x256')):
        res = {}
        for h3 in (False, True):
            t = timed(lambda: L.gemm(*args, nb=(1, nb), tile=tile, splitk=sk, workspace=ws, f16x3=h3))
            err = ((C[0][rows].double() - ref).abs() / mag).max().item()
            res[h3] = (t, err)
        print('    %-10s bf16x6 %7.3f ms %6.1f TF err %.2e | f16x3 %7.3f ms %6.1f TF err %.2e | %+.1f %%' % (
            tn, res[False][0], fl / res[False][0] / 1e9, res[False][1], res[True][0], fl / res[True][0] / 1e9, res[True][1],
            100 * (res[False][0] / res[True][0] - 1)), flush=True)
