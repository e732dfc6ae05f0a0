"""Where the time of the short-contraction attention GEMMs goes (a measured statement of the bound): the 4-wave bf16x6 kernel on the
K = 256 P.V shapes of cfg2 / cfg4 / cfg5 and, for contrast, a long contraction -- product schedule against the ablations of a -DSEGX_BENCH build (knob 6: 2 = no split
arithmetic, 3 = no LDS stores, 4 = no global loads after the first k-tile, 5 = matrix instructions + fragment reads + epilogue only), random and all-zero operands.
GPU box:  python tools/build_variant.py bench gemm.hip -DSEGX_BENCH && python tools/attn_bound.py"""
import glob, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx
dev = torch.device('cuda', 0)
path = os.path.join(ROOT, 'tools', 'variants', 'libsegx_bench.so')
L = segx.SegxLib(path if os.path.exists(path) else segx.LIB_PATH)
bench_build = os.path.exists(path)
L.set_engine('x6')
# name, M, N, K, nb (A [nb, M, K] and B [nb, N, K] both k-contiguous: the NT layout the ablation variants are built for)
SHAPES = [('cfg2 P.V        ', 4096, 1792, 256, 24), ('cfg2 QK^T (NT) ', 4096, 256, 1792, 24), ('cfg4 P.V        ', 2352, 1024, 256, 16), ('cfg5 P.V        ', 4096, 1024, 256, 16),
          ('cfg2 K = 512    ', 4096, 1792, 512, 24), ('cfg2 K = 1792   ', 4096, 1792, 1792, 24)]
NAMES = {0: 'product', 2: 'no split math', 3: 'no LDS stores', 4: 'no global loads', 5: 'MFMA + frag reads + epilogue'}


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


g = torch.Generator(device='cpu').manual_seed(0)
print('bench build: %s' % bench_build)
for name, M, N, K, nb in SHAPES:
    flops = 2.0 * M * N * K * nb
    for data in ('random', 'zeros'):
        A = (torch.randn(nb, M, K, generator=g) if data == 'random' else torch.zeros(nb, M, K)).to(dev)
        B = (torch.randn(nb, N, K, generator=g) if data == 'random' else torch.zeros(nb, N, K)).to(dev)
        C = torch.empty(nb, M, N, device=dev)
        row = '%s %-6s' % (name, data)
        for var in ((0, 2, 3, 4, 5) if bench_build else (0,)):
            if L.c.segx_tune(6, var) != 0:
                continue
            t = timed(lambda: L.gemm(A, B, C, M, N, K, (M * K, 0, K, 1), (N * K, 0, K, 1), (M * N, 0, N), nb=(nb, 1), tile=segx.TILE_128x128))
            row += ' | %s %6.3f ms %6.1f TF' % (NAMES[var], t, flops / t / 1e9)
        L.c.segx_tune(6, 0)
        print(row, flush=True)
