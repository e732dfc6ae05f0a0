"""Within-process A/B of the bf16x6 GEMM kernels on the shapes of the cfg-2 / cfg-4 steps (GPU box):
python tools/ws_bench.py [rounds] [reps] [shape-set]
Arms = (library, tile, knob-6 variant); every arm runs once per round, rounds interleaved, median / min over rounds from HIP events
(cdna_hip_programming.md 5.4 rule 24).  Libraries: the product build + every tools/variants/libsegx_*.so (tools/build_variant.py)."""
import glob, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

dev = torch.device('cuda', 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
which = sys.argv[3] if len(sys.argv) > 3 else 'main'
LIBS = {'prod': segx.SegxLib(segx.LIB_PATH)}
for p in sorted(glob.glob(os.path.join(ROOT, 'tools', 'variants', 'libsegx_*.so'))):
    LIBS[os.path.basename(p)[8:-3]] = segx.SegxLib(p)
for L in LIBS.values():
    L.set_engine('x6')
g = torch.Generator(device='cpu').manual_seed(0)
TN = {0: 'auto', 1: '128x128', 2: '64x64', 5: '64x128', 6: 'ws256x128', 7: 'ws128x128'}

# name, M, N, K, akc, bkc, nb, splitk (0 = planned), gelu
SHAPES = {
    'main': [('group_linear fwd NT', 24576, 1792, 1792, True, True, 4, 1, False), ('group_linear dX NN', 24576, 1792, 1792, True, False, 4, 1, False),
             ('group_linear dW TN sk3', 1792, 1792, 24576, False, False, 4, 3, False), ('scores QK^T', 4096, 256, 1792, True, False, 24, 1, False),
             ('scores QK^T NT', 4096, 256, 1792, True, True, 24, 1, False), ('P.V', 4096, 1792, 256, True, True, 24, 1, False),
             ('fusion GELU P.(vW)', 4096, 1792, 256, True, False, 24, 1, True), ('dP 1792x256x4096', 1792, 256, 4096, False, False, 24, 1, False),
             ('l2 linear 896', 24576, 896, 896, True, True, 4, 1, False), ('l3 linear 448', 24576, 448, 448, True, True, 4, 1, False),
             ('attn small 4096x896x256', 4096, 896, 256, True, False, 24, 1, False), ('square 8192', 8192, 8192, 8192, True, True, 1, 1, False)],
    'backbone': [('pw 960->160', 160, 4096, 960, True, False, 6, 1, False), ('pw 160->960', 960, 4096, 160, True, False, 6, 1, False),
                 ('pw 1632->272', 272, 1024, 1632, True, False, 6, 1, False), ('pw 272->1632', 1632, 1024, 272, True, False, 6, 1, False),
                 ('pw 448->1792', 1792, 1024, 448, True, False, 6, 1, False), ('head 6144x1792x1792', 6144, 1792, 1792, True, True, 1, 1, False),
                 ('3d fpn 192->480', 480, 150528, 192, True, False, 4, 1, False), ('3d outfpn 832 comp', 832, 37632, 480, True, False, 4, 1, False)],
    'lean': [('pw 960->160', 160, 4096, 960, True, False, 6, 1, False), ('pw 160->960', 960, 4096, 160, True, False, 6, 1, False),
             ('pw 1632->272', 272, 1024, 1632, True, False, 6, 1, False), ('pw 448->1792', 1792, 1024, 448, True, False, 6, 1, False),
             ('3d fpn 192->480', 480, 150528, 192, True, False, 4, 1, False), ('3d outfpn 832 comp', 832, 37632, 480, True, False, 4, 1, False),
             ('P.V', 4096, 1792, 256, True, True, 24, 1, False), ('attn small 4096x896x256', 4096, 896, 256, True, False, 24, 1, False),
             ('l3 linear 448', 24576, 448, 448, True, True, 4, 1, False), ('dW small TN', 448, 448, 24576, False, False, 4, 8, False),
             ('scores QK^T', 4096, 256, 1792, True, False, 24, 1, False), ('fusion GELU P.(vW)', 4096, 1792, 256, True, False, 24, 1, True)],
    'k256': [('P.V', 4096, 1792, 256, True, True, 24, 1, False), ('fusion GELU P.(vW)', 4096, 1792, 256, True, False, 24, 1, True),
             ('attn small 4096x896x256', 4096, 896, 256, True, False, 24, 1, False), ('K=512', 4096, 1792, 512, True, True, 24, 1, False),
             ('scores QK^T', 4096, 256, 1792, True, False, 24, 1, False)],
    'ablate': [('group_linear fwd NT', 24576, 1792, 1792, True, True, 4, 1, False), ('scores QK^T NT', 4096, 256, 1792, True, True, 24, 1, False),
               ('P.V', 4096, 1792, 256, True, True, 24, 1, False), ('square 8192', 8192, 8192, 8192, True, True, 1, 1, False)],
    'pmc': [('group_linear fwd NT', 24576, 1792, 1792, True, True, 4, 1, False), ('scores QK^T', 4096, 256, 1792, True, False, 24, 1, False),
            ('P.V', 4096, 1792, 256, True, True, 24, 1, False)],
}
if which == 'pmc':
    LIBS = {'prod': LIBS['prod']}
    PMC_ARMS = [('prod', 1, 0), ('prod', 6, 0), ('prod', 6, 5), ('prod', 7, 0), ('prod', 7, 5)]
ARMS = [('prod', 1, 0), ('prod', 6, 0), ('prod', 7, 0)] + [(n, 6, 0) for n in LIBS if n != 'prod']
if which == 'pmc':
    ARMS = PMC_ARMS
if which == 'lean':                                       # 4-wave kernels: lean loaders (prod) against the per-k-tile address form (variant 'nolean')
    ARMS = [(n, t, 0) for t in (1, 5, 2) for n in ('prod', 'nolean') if n in LIBS]
if which == 'k256':                                       # short contractions: product tiles against any variant builds present
    ARMS = [('prod', 1, 0), ('prod', 6, 0), ('prod', 7, 0)] + [(n, t, 0) for n in LIBS if n != 'prod' for t in (1, 6)]
if which == 'ablate':                                     # knob 6 on the wave-specialised kernel: 2 no split math, 3 no global loads, 4 no LDS stores, 5 consumers alone
    ARMS = [('prod', 6, v) for v in (0, 1, 2, 3, 4, 5)] + [('prod', 7, v) for v in (0, 2, 5)]


def make(M, N, K, akc, bkc, nb, sk, gelu):
    A = torch.randn(nb, M, K, generator=g).to(dev) if akc else torch.randn(nb, K, M, generator=g).to(dev)
    B = torch.randn(nb, N, K, generator=g).to(dev) if bkc else torch.randn(nb, K, N, generator=g).to(dev)
    C = torch.empty(nb, M, N, device=dev)
    a = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    kw = dict(nb=(1, nb), splitk=sk, workspace=torch.empty(sk * nb * M * N, device=dev) if sk > 1 else None)
    if gelu:
        kw.update(epilogue=segx.EPI_GELU, aux=torch.empty(nb, M, N, device=dev), dropout_p=0.2, seed=1, offset=0,
                  bias=torch.randn(nb, N, generator=g).to(dev), bias_mode=segx.BIAS_N, bias_b1=N)
    return (A, B, C, M, N, K, a, b, (0, M * N, N)), kw


for sh in SHAPES[which]:
    name, M, N, K, akc, bkc, nb, sk, gelu = sh
    args, kw = make(*sh[1:])
    times = {arm: [] for arm in ARMS}
    for arm in ARMS:                                       # warm-up, and which arms this shape supports
        L = LIBS[arm[0]]
        L.c.segx_tune(6, arm[2])
        try:
            L.gemm(*args, tile=arm[1], **kw)
        except RuntimeError as e:
            times.pop(arm); print('  skip', arm, str(e)[:80])
        L.c.segx_tune(6, 0)
    torch.cuda.synchronize()
    if which == 'main':                                    # every library computes the same values (the split variants are exact re-expressions)
        ref = None
        for n, L in LIBS.items():
            args[2].fill_(float('nan')); L.gemm(*args, tile=6, **kw); torch.cuda.synchronize()
            if ref is None: ref = args[2].clone()
            elif not torch.equal(ref, args[2]): print('  !! library %s differs from prod: max |d| = %.3e' % (n, (ref - args[2]).abs().max().item()))
    for r in range(rounds):
        for arm in list(times):
            L = LIBS[arm[0]]
            L.c.segx_tune(6, arm[2])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.gemm(*args, tile=arm[1], **kw)
            e1.record(); torch.cuda.synchronize()
            times[arm].append(e0.elapsed_time(e1) / reps)
            L.c.segx_tune(6, 0)
    fl = 2.0 * M * N * K * nb
    print('%-26s M=%6d N=%6d K=%6d nb=%2d sk=%d %s%s' % (name, M, N, K, nb, sk, 'NT'[0] if akc else 'T', 'T' if bkc else 'N'), flush=True)
    for arm, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print('    %-8s %-10s v%d  median %7.3f ms %6.1f TF   best %7.3f ms %6.1f TF   frac(416.7) %.3f' % (arm[0], TN[arm[1]], arm[2], med, fl / med / 1e9, mn, fl / mn / 1e9,
                                                                                                          fl / med / 1e9 / 416.7), flush=True)
