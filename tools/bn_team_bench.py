"""Training BatchNorm on the planes that do not fit one workgroup: the team form (segx_tune knob 3 = 0) against the two-launch form (knob 3 = 1), forward and
backward, HIP events, on the three large plane sizes of EfficientNet-B4 at 512 x 512 / batch 6; on the small ones the channel-resident form (default) against teams (knob 3 = 2).  python tools/bn_team_bench.py [libsegx variant .so]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx
L = segx.SegxLib(sys.argv[1]) if len(sys.argv) > 1 else segx.lib()
dev = torch.device('cuda', 0)
B = 6


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C, S in ((336, 16384), (192, 65536), (144, 262144), (960, 4096), (672, 4096), (1632, 1024)):
    x = torch.randn(B, C, S, device=dev); dy = torch.randn(B, C, S, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    mean, var = torch.empty(C, device=dev), torch.empty(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dw, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    mb = B * C * S * 4 / 1e6
    for path, name in (((1, 'two launches'), (0, 'team')) if S >= 16384 else ((0, 'resident'), (2, 'team'))):
        assert L.c.segx_tune(3, path) == 0
        parts = torch.empty(L.bn_parts_floats(B, C, S), device=dev); ws = torch.empty(L.bn_ws(B, C, S), device=dev)     # sized under the form they serve
        tf = timeit(lambda: L.bn_act_fwd2(x, parts, 0, mean, var, rm, rv, 0.01, w, b, y, None, None, 0.0, 0, 0, B, C, S, 1e-3, 1))
        tb = timeit(lambda: L.bn_act_bwd2(dy, x, mean, var, w, b, dx, dw, db, ws, B, C, S, 1e-3, 1, 1))
        print('C %4d S %6d %-12s fwd %7.1f us (%5.2f TB/s of 2 units)  bwd %7.1f us (%5.2f TB/s of 3 units)' % (C, S, name, tf, 2 * mb / tf, tb, 3 * mb / tb), flush=True)
    L.c.segx_tune(3, 0)
