"""Micro-benchmark of the two tile engines (fp32 MFMA / bf16x6) on the shapes that dominate the cfg-2 / cfg-4 train steps.
python tools/gemm_bench.py [reps] [tiles]   (GPU box).  Prints TFLOP/s (fp32-equivalent) per shape, engine and tile from HIP events;
`tiles` sweeps every tile of both engines over the backbone / attention shapes (input of the planners' cost-model constants)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segtran_amd import segx

L = segx.lib()
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device='cpu').manual_seed(0)
NAMES = {0: 'auto', 1: '128x128', 2: '64x64', 3: '128x32', 4: '32x128', 5: '64x128', 6: '256x128'}


def run(name, M, N, K, akc, bkc, splitk=1, nb=1, engine='x6', tile=0):
    L.set_engine(engine)
    A = torch.randn(nb, M, K, generator=g).to(dev) if akc else torch.randn(nb, K, M, generator=g).to(dev)
    B = torch.randn(nb, N, K, generator=g).to(dev) if bkc else torch.randn(nb, K, N, generator=g).to(dev)
    C = torch.empty(nb, M, N, device=dev)
    a = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    c = (0, M * N, N)
    kw = dict(nb=(1, nb), tile=tile)
    if splitk == 0:
        kw['splitk'] = 0                                   # library-planned tile + split factor
    else:
        kw.update(splitk=splitk, workspace=torch.empty(splitk * nb * M * N, device=dev) if splitk > 1 else None)
    L.x6_launches()
    for _ in range(2):
        L.gemm(A, B, C, M, N, K, a, b, c, **kw)
    on6 = L.x6_launches() > 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.gemm(A, B, C, M, N, K, a, b, c, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%-30s M=%6d N=%6d K=%6d nb=%2d sk=%2d %-4s %-8s %8.3f ms  %6.1f TFLOP/s' % (name, M, N, K, nb, splitk, 'x6' if on6 else 'f32', NAMES[tile], ms,
                                                                                      2.0 * M * N * K * nb / ms / 1e9), flush=True)


MAIN = [('group_linear fwd NT x4', 24576, 1792, 1792, True, True, 1, 4), ('group_linear dX NN x4', 24576, 1792, 1792, True, False, 1, 4),
        ('group_linear dW TN x4', 1792, 1792, 24576, False, False, 0, 4), ('fusion GELU-shape 4096x1792x256 x24', 4096, 1792, 256, True, False, 1, 24),
        ('scores 4096x256x1792 x24', 4096, 256, 1792, True, False, 1, 24), ('PV dP 1792x256x4096 x24', 1792, 256, 4096, False, False, 1, 24),
        ('l2 linear 24576x896x896 x4', 24576, 896, 896, True, True, 1, 4), ('head 1792->1792 6144 rows', 6144, 1792, 1792, True, True, 1, 1),
        ('square 4096 NT', 4096, 4096, 4096, True, True, 1, 1), ('square 8192 NT', 8192, 8192, 8192, True, True, 1, 1),
        ('pw fwd 960->160', 160, 4096, 960, True, False, 1, 6), ('pw fwd 160->960', 960, 4096, 160, True, False, 1, 6),
        ('pw fwd 1632->272', 272, 1024, 1632, True, False, 1, 6), ('pw fwd 272->1632', 1632, 1024, 272, True, False, 1, 6),
        ('pw dgrad 160->960', 160, 4096, 960, False, False, 1, 6), ('pw wgrad 960x160', 960, 160, 4096, True, True, 0, 6),
        ('pw fwd 56->336', 336, 16384, 56, True, False, 1, 6), ('pw fwd 448->1792 head', 1792, 1024, 448, True, False, 1, 6),
        ('3d outfpn 832->4 comp', 832, 37632, 480, True, False, 1, 4), ('3d fpn 192->480', 480, 150528, 192, True, False, 1, 4)]

if len(sys.argv) > 2 and sys.argv[2] == 'variants':
    # what the parts of the 128 x 128 k-tile loop cost (segx_tune knob 6; variants >= 2 do NOT compute the GEMM), and the 256 x 128 tile
    VN = {0: 'product', 1: 'setprio', 2: 'no split math', 3: 'no LDS stores', 4: 'no global loads', 5: 'MFMA + frag reads only'}
    for sh in (MAIN[0], MAIN[8], MAIN[9], MAIN[6]):
        for v in range(6):
            L.c.segx_tune(6, v)
            run('%s [%s]' % (sh[0][:14], VN[v]), *sh[1:7], nb=sh[7], engine='x6', tile=1)
        L.c.segx_tune(6, 0)
    for sh in MAIN[:10] + [('expand 112->672', 672, 4096, 112, True, False, 1, 6)]:
        for tile in (1, 6, 5, 2):
            run(sh[0], *sh[1:7], nb=sh[7], engine='x6', tile=tile)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == 'tiles':
    for sh in MAIN:
        run(sh[0], *sh[1:7], nb=sh[7], engine='f32', tile=0)
        for tile in (1, 5, 2):
            run(sh[0], *sh[1:7], nb=sh[7], engine='x6', tile=tile)
    sys.exit(0)
for sh in MAIN:
    for eng in ('f32', 'x6'):
        run(sh[0], *sh[1:7], nb=sh[7], engine=eng)
