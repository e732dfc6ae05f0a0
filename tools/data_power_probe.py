"""Does the bf16x6 GEMM's speed depend on the operand VALUES?  (python tools/data_power_probe.py)
Same kernel, same shape (24576 x 1792 x 1792 x 4, wave-specialised 256 x 128): normal random operands, all zeros, one repeated value, small integers
(mid / lo planes zero).  A difference is the chip's power management (clock under dense bf16 MFMA load), not the schedule."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx
dev = torch.device('cuda', 0)
L = segx.lib(); L.set_engine('x6')
M, N, K, nb = 24576, 1792, 1792, 4
g = torch.Generator(device='cpu').manual_seed(0)
C = torch.empty(nb, M, N, device=dev)


def timed(A, B, tile, reps=8, rounds=5):
    fn = lambda: L.gemm(A, B, C, M, N, K, (0, M * K, K, 1), (0, N * K, K, 1), (0, M * N, N), nb=(1, nb), tile=tile)
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


cases = {'randn': (torch.randn(nb, M, K, generator=g).to(dev), torch.randn(nb, N, K, generator=g).to(dev)),
         'zeros': (torch.zeros(nb, M, K, device=dev), torch.zeros(nb, N, K, device=dev)),
         'constant 1.2345678': (torch.full((nb, M, K), 1.2345678, device=dev), torch.full((nb, N, K), 1.2345678, device=dev)),
         'small integers (mid = lo = 0)': (torch.randint(-3, 4, (nb, M, K), generator=g).float().to(dev), torch.randint(-3, 4, (nb, N, K), generator=g).float().to(dev))}
fl = 2.0 * M * N * K * nb
for tile, tn in ((segx.TILE_256x128, 'ws256x128'), (segx.TILE_128x128, '4-wave 128x128')):
    for name, (A, B) in cases.items():
        t = timed(A, B, tile)
        print('%-16s %-32s %7.3f ms  %6.1f TFLOP/s' % (tn, name, t, fl / t / 1e9), flush=True)
