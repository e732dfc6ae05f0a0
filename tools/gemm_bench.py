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


run('linear fwd NT', 24576, 1792, 1792, True, True)
run('linear fwd NT (x4 modes)', 98304, 1792, 1792, True, True)
run('dX NN', 24576, 1792, 1792, True, False)
run('dW TN splitk5', 1792, 1792, 24576, False, False, splitk=5)
run('square 4096 NT', 4096, 4096, 4096, True, True)
run('square 8192 NT (16 rounds)', 8192, 8192, 8192, True, True)
run('2048 tiles exactly (4 rounds)', 16384, 2048, 1792, True, True)
run('512 tiles exactly (1 round)', 8192, 1024, 4096, True, True)
run('256 tiles (half round)', 4096, 1024, 4096, True, True)
