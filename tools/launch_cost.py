"""Host cost of ONE launch through the binding, against an ATen launch of the same size: the eager step's floor is launches x this.  GPU box: python tools/launch_cost.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx, functional as SF

dev = torch.device('cuda', 0)
L = segx.lib()
L.set_engine('x6')
N = 3000
x = torch.randn(8, 64, device=dev); out = torch.empty(64, device=dev); ws = torch.empty(max(1, L.colreduce_ws(8, 64, 1)), device=dev)
A = torch.randn(64, 64, device=dev); Bm = torch.randn(64, 64, device=dev); C = torch.empty(64, 64, device=dev)


def timeit(name, fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-58s %6.2f us issue   %6.2f us incl. drain' % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))


raw = L.c.segx_colsum
st = L.stream(x)
px, po, pw = x.data_ptr(), out.data_ptr(), ws.data_ptr()
timeit('aten add_ (tiny)', lambda: out.add_(1.0))
timeit('torch.empty(64)', lambda: torch.empty(64, device=dev))
timeit('raw ctypes segx_colsum (pointers ready)', lambda: raw(px, po, pw, 8, 64, st))
timeit('L.colsum (binding: checks + pointers + stream)', lambda: L.colsum(x, out, ws, 8, 64))
timeit('L.gemm 64^3 (descriptor + plan + launch)', lambda: L.gemm(A, Bm, C, 64, 64, 64, (0, 0, 64, 1), (0, 0, 64, 1), (0, 0, 64), splitk=0))
timeit('SF.linear 64^3 no grad (op wrapper)', lambda: SF.linear(A, Bm))
Ag = A.clone().requires_grad_(True)
timeit('SF.linear 64^3 with autograd node', lambda: SF.linear(Ag, Bm))
timeit('L.stream()', lambda: L.stream(x))
