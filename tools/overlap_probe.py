"""Do an MFMA-bound tile-engine GEMM and an HBM-bound streaming kernel overlap when launched on two HIP streams?  (python tools/overlap_probe.py)
Prints the time of each alone, back to back on one stream, and concurrently on two streams."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx, functional as SF

dev = torch.device('cuda', 0)
L = segx.lib()
g = torch.Generator().manual_seed(0)
A = torch.randn(4, 24576, 1792, generator=g).to(dev)              # dY of a mode-batched projection
W = torch.randn(4, 1792, 1792, generator=g).to(dev)
C = torch.empty(4, 24576, 1792, device=dev)
x = torch.randn(6, 144, 128, 128, generator=g).to(dev)           # an expanded MBConv activation (57 MB)
bn = torch.nn.BatchNorm2d(144).to(dev)
side = torch.cuda.Stream()


def gemm():
    L.gemm(A, W, C, 24576, 1792, 1792, (24576 * 1792, 0, 1792, 1), (1792 * 1792, 0, 1792, 1), (24576 * 1792, 0, 1792), nb=(4, 1), splitk=0)


def glue(n=96):
    for _ in range(n):
        SF.bn_act(x, bn, SF.ACT_SWISH)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def both_serial():
    gemm(); glue()


def both_overlap():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gemm()
    glue()
    torch.cuda.current_stream().wait_stream(side)


with torch.no_grad():
    print('gemm alone %.3f ms   glue alone %.3f ms   one stream %.3f ms   two streams %.3f ms' % (timed(gemm), timed(glue), timed(both_serial), timed(both_overlap)))
