"""Streaming skinny weight gradient (gemm_skinny.hip, knob 18) against the tile kernels' split-K slabs on the batch-reduced weight-gradient shapes of cfg2's
first pointwise convolutions: python tools/skinny_bench.py [reps]   (HIP events around segx_gemm_f32 incl. its slab reduction; algorithmic bytes / time)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = segx.lib()
dev = torch.device('cuda', 0)
SHAPES = [(192, 32, 65536, 6), (144, 24, 262144, 6), (32, 192, 65536, 6), (24, 48, 262144, 6), (24, 24, 262144, 6), (32, 144, 65536, 6), (3, 160, 65536, 6),
          (56, 32, 65536, 6), (160, 32, 16384, 6), (4, 64, 150528, 4)]
print('%-28s %10s %10s %8s %8s' % ('M N K batch', 'tiles us', 'stream us', 'TB/s', 'TB/s'))
for M, N, K, nb in SHAPES:
    A = torch.randn(nb, M, K, device=dev); B = torch.randn(nb, N, K, device=dev); C = torch.empty(M, N, device=dev)
    t = {}
    for knob in (0, 1):
        assert L.c.segx_tune(18, knob) == 0
        for _ in range(3):
            L.gemm(A, B, C, M, N, K, (M * K, 0, K, 1), (N * K, 0, K, 1), (0, 0, N), nb=(nb, 1), splitk=0, batch_reduce=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.gemm(A, B, C, M, N, K, (M * K, 0, K, 1), (N * K, 0, K, 1), (0, 0, N), nb=(nb, 1), splitk=0, batch_reduce=True)
        e1.record(); torch.cuda.synchronize()
        t[knob] = e0.elapsed_time(e1) / reps * 1e3
        if knob == 0:
            ref = C.clone()
    err = (C - ref).abs().max().item() / max(ref.abs().max().item(), 1e-20)
    by = 4.0 * (M + N) * K * nb
    print('%-28s %10.1f %10.1f %8.2f %8.2f   rel diff %.1e' % ((M, N, K, nb), t[0], t[1], by / t[0] / 1e6, by / t[1] / 1e6, err))
L.c.segx_tune(18, 1)
