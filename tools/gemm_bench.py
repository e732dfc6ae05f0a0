"""Micro-benchmark of the fp32-MFMA GEMM engine on the shapes that dominate the cfg-2 train step.
python tools/gemm_bench.py [reps]   (GPU box).  Prints TFLOP/s per shape from HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segtran_amd import segx

L = segx.lib()
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device='cpu').manual_seed(0)


def run(name, M, N, K, akc, bkc, splitk=1, nb=1):
    A = torch.randn(nb, M, K, generator=g).to(dev) if akc else torch.randn(nb, K, M, generator=g).to(dev)
    B = torch.randn(nb, N, K, generator=g).to(dev) if bkc else torch.randn(nb, K, N, generator=g).to(dev)
    C = torch.empty(nb, M, N, device=dev)
    ws = torch.empty(splitk * nb * M * N, device=dev) if splitk > 1 else None
    a = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    c = (0, M * N, N)
    for _ in range(2):
        L.gemm(A, B, C, M, N, K, a, b, c, nb=(1, nb), splitk=splitk, workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.gemm(A, B, C, M, N, K, a, b, c, nb=(1, nb), splitk=splitk, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%-34s M=%6d N=%5d K=%6d nb=%d sk=%2d  %8.3f ms  %6.1f TFLOP/s' % (name, M, N, K, nb, splitk, ms, 2.0 * M * N * K * nb / ms / 1e9))


if len(sys.argv) > 2 and sys.argv[2] == 'tiles':
    # tile sweep over the pointwise-convolution shapes of the EfficientNet-B4 backbone at cfg 2 (bs 6, 512 x 512)
    SHAPES = [('pw wgrad 192x32', 192, 32, 65536, True, True, 32, 6), ('pw wgrad 144x24', 144, 24, 262144, True, True, 32, 6),
              ('pw wgrad 32x192', 32, 192, 65536, True, True, 32, 6), ('pw wgrad 24x48', 24, 48, 262144, True, True, 32, 6),
              ('pw wgrad 336x56', 336, 56, 16384, True, True, 27, 6), ('pw wgrad 448x160', 448, 160, 65536, True, True, 10, 6),
              ('pw wgrad 960x160', 960, 160, 4096, True, True, 5, 6), ('pw wgrad 1632x272', 1632, 272, 1024, True, True, 2, 6),
              ('pw fwd 32->192', 192, 65536, 32, True, False, 1, 6), ('pw fwd 192->32', 32, 65536, 192, True, False, 1, 6),
              ('pw dgrad 192->32', 32, 65536, 192, False, False, 1, 6), ('pw fwd 24->144', 144, 262144, 24, True, False, 1, 6),
              ('pw dgrad 144->24', 24, 262144, 144, False, False, 1, 6), ('pw fwd 56->336', 336, 16384, 56, True, False, 1, 6),
              ('pw dgrad 336->56', 56, 16384, 336, False, False, 1, 6), ('pw fwd 960->160', 160, 4096, 960, True, False, 1, 6),
              ('pw dgrad 160->960', 160, 4096, 960, False, False, 1, 6), ('pw fwd 160->960', 960, 4096, 160, True, False, 1, 6),
              ('pw fwd 1632->272', 272, 1024, 1632, True, False, 1, 6), ('pw fwd 272->1632', 1632, 1024, 272, True, False, 1, 6),
              ('pw fwd 672->112', 112, 4096, 672, True, False, 1, 6), ('pw fwd 112->672', 672, 4096, 112, True, False, 1, 6),
              ('infpn 448x65536x160', 448, 65536, 160, True, False, 1, 6), ('scores 256x4096x1792', 256, 4096, 1792, True, True, 1, 6),
              ('PV 4096x1792x256', 4096, 1792, 256, False, False, 1, 6), ('linear 24576x1792', 24576, 1792, 1792, True, True, 1, 1)]
    names = {0: 'auto', 1: '128x128', 2: '64x64', 3: '128x32', 4: '32x128', 5: '64x128'}
    for sh in SHAPES:
        for tile in range(6):
            L.force_tile = tile
            run('%s [%s]' % (sh[0], names[tile]), sh[1], sh[2], sh[3], sh[4], sh[5], splitk=sh[6], nb=sh[7])
    sys.exit(0)

run('linear fwd NT', 24576, 1792, 1792, True, True)
run('linear fwd NT (x4 modes)', 98304, 1792, 1792, True, True)
run('dX NN', 24576, 1792, 1792, True, False)
run('dW TN splitk5', 1792, 1792, 24576, False, False, splitk=5)
run('square 4096 NT', 4096, 4096, 4096, True, True)
run('square 8192 NT (16 rounds)', 8192, 8192, 8192, True, True)
run('2048 tiles exactly (4 rounds)', 16384, 2048, 1792, True, True)
run('512 tiles exactly (1 round)', 8192, 1024, 4096, True, True)
run('256 tiles (half round)', 4096, 1024, 4096, True, True)
