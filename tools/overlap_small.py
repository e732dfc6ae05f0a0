"""Do the two backward GEMMs of a backbone pointwise convolution (dX and dW: independent, each too small to fill 256 CUs for long) overlap when
launched on two HIP streams?   python tools/overlap_small.py
Per (Cin, Cout, HW) of EfficientNet-B4 at 512^2, batch 6: time of the pair back to back on one stream and forked / joined over two streams
(the fork and join are in the timed loop, as they would be in a train step)."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

dev = torch.device('cuda', 0)
L = segx.lib()
L.set_engine('x6')
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
B = 6
CASES = [(1632, 272, 1024), (272, 1632, 1024), (960, 160, 4096), (160, 960, 4096), (672, 112, 4096), (112, 672, 4096), (336, 56, 16384), (56, 336, 16384),
         (448, 1792, 1024), (1632, 448, 256), (2688, 448, 256)]


def plan(A, Bm, M, N, K, a, b, c, nb, batch_reduce=False):
    """tile, split factor and workspace once (what SegxLib.gemm(splitk=0) does per call)."""
    import ctypes
    d = segx.GemmDesc(); d.M, d.N, d.K, d.nb0, d.nb1 = M, N, K, nb[0], nb[1]
    d.a_b0, d.a_b1, d.a_m, d.a_k = a; d.b_b0, d.b_b1, d.b_n, d.b_k = b; d.c_b0, d.c_b1, d.c_m = c; d.alpha = 1.0
    d.batch_reduce = 1 if batch_reduce else 0
    t, sk = ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.c.segx_gemm_plan(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(Bm.data_ptr()), ctypes.byref(d), ctypes.byref(t), ctypes.byref(sk)), 'plan')
    ws = torch.empty(sk.value * nb[0] * nb[1] * M * N, device=dev) if (sk.value > 1 or batch_reduce) else None
    return t.value, sk.value, ws


for Cin, Cout, HW in CASES:
    W = torch.randn(Cout, Cin, generator=g).to(dev)
    X = torch.randn(B, Cin, HW, generator=g).to(dev)
    dY = torch.randn(B, Cout, HW, generator=g).to(dev)
    dX = torch.empty(B, Cin, HW, device=dev); dW = torch.empty(Cout, Cin, device=dev)
    # dX[b] = W^T dY[b]: M = Cin, N = HW, K = Cout; A = W^T (m stride 1, k stride Cin), B = dY[b] (n stride 1, k stride HW)
    ax, bx, cx = (0, 0, 1, Cin), (0, Cout * HW, 1, HW), (0, Cin * HW, HW)
    tx, skx, wsx = plan(W, dY, Cin, HW, Cout, ax, bx, cx, (1, B))
    # dW = sum_b dY[b] X[b]^T: M = Cout, N = Cin, K = HW, both k-contiguous, reduced over the batch
    aw, bw, cw = (0, Cout * HW, HW, 1), (0, Cin * HW, HW, 1), (0, 0, Cin)
    tw, skw, wsw = plan(dY, X, Cout, Cin, HW, aw, bw, cw, (1, B), batch_reduce=True)

    def g_dx():
        L.gemm(W, dY, dX, Cin, HW, Cout, ax, bx, cx, nb=(1, B), splitk=skx, workspace=wsx, tile=tx)

    def g_dw():
        L.gemm(dY, X, dW, Cout, Cin, HW, aw, bw, cw, nb=(1, B), splitk=skw, workspace=wsw, tile=tw, batch_reduce=True)

    def serial():
        g_dx(); g_dw()

    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g_dw()
        g_dx()
        cur.wait_stream(side)

    def timed(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        return statistics.median(ts)

    a, b, s, f = timed(g_dx), timed(g_dw), timed(serial), timed(forked)
    print('Cin %5d Cout %5d HW %6d  dX %6.1f us (tile %d sk %d)  dW %6.1f us (tile %d sk %d)  one stream %6.1f us  two streams %6.1f us  gain %.2fx' % (
        Cin, Cout, HW, a, tx, skx, b, tw, skw, s, f, s / f), flush=True)
