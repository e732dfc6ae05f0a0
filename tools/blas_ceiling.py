"""What does the vendor fp32 GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on the shapes that dominate cfg 2?  A ceiling check for
the hand-written engine, not a product path.  python tools/blas_ceiling.py   (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segtran_amd import segx
torch.backends.cuda.matmul.allow_tf32 = False
L = segx.lib(); dev = torch.device('cuda', 0)
for (M, N, K) in [(24576, 1792, 1792), (98304, 1792, 1792), (1792, 1792, 24576), (24576, 896, 896), (4096, 4096, 4096), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    def t(fn, reps=10):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    tb = t(lambda: torch.matmul(A, B.t(), out=C))
    ts = t(lambda: L.gemm(A, B, C, M, N, K, (0, 0, K, 1), (0, 0, K, 1), (0, 0, N), splitk=0))
    f = 2.0 * M * N * K / 1e9
    print('M=%6d N=%5d K=%6d  vendor %7.3f ms %6.1f TF   segx %7.3f ms %6.1f TF' % (M, N, K, tb, f / tb, ts, f / ts))
