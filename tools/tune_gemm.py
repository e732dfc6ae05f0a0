"""Measured tile / split-K choices for the GEMM shapes of the BASELINE configurations (GPU box):
    python tools/tune_gemm.py <shapes.txt> [<shapes.txt> ...] > gpurun_out/<tag>/tune_gemm.txt
reads the `[bench] (M, N, K, batch, A_kcontig, B_kcontig, splitk, tile, 'x6')` lines bench.py prints under SEGX_BENCH_VERBOSE=2, times every
bf16x6 tile (4-wave 128x128 / 64x128 / 64x64, wave-specialised 256x128 / 128x128 / 128x256 / 64x256 / 96x256 / 256x96) x a few split-K factors per unique shape with HIP events,
and prints one line per shape: the cost model's choice (segx_gemm_plan_model: the table of the previous sweep is NOT consulted) and its time, the best choice and time.  tools/tune_table.py turns the output into
segtran_amd/csrc/gemm_tuned.h (entries where the measured best beats the cost model's pick by > 4 %), which segx_gemm_plan consults first.
The cost model stays the fallback for every shape that is not in the table."""
import os, re, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx

L = segx.lib()
L.set_engine('x6')
dev = torch.device('cuda', 0)
g = torch.Generator(device='cpu').manual_seed(0)
shapes = {}
for path in sys.argv[1:]:
    for l in open(path):
        m = re.match(r"\[bench\]\s+\((\d+), (\d+), (\d+), (\d+), (True|False), (True|False), (\d+), (\d+), 'x6'\)\s+(\d+)\s+([\d.]+)", l)
        if m:
            M, N, K, nb = (int(m.group(i)) for i in (1, 2, 3, 4))
            key = (M, N, K, nb, m.group(5) == 'True', m.group(6) == 'True')
            shapes[key] = shapes.get(key, 0.0) + float(m.group(10))
print('# %d unique bf16x6 GEMM shapes' % len(shapes), flush=True)
BUF = {}


def buf(n):
    if n not in BUF:
        BUF[n] = torch.randn(n, generator=g).to(dev)
    return BUF[n]


def setup(M, N, K, nb, akc, bkc, sk, C, ws):
    A, B = buf(nb * M * K), buf(nb * N * K)
    a = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    b = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    return (A, B, C, M, N, K, a, b, (0, M * N, N)), dict(nb=(1, nb), splitk=sk, workspace=ws if sk > 1 else None)


def burst(args, kw, tile, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.gemm(*args, tile=tile, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sweep(M, N, K, nb, akc, bkc, cands, rounds=3):
    """Median time of every (tile, split-K) candidate: the device is first driven for ~30 ms with the first candidate (the buffers were just generated on the
    host: an idle device clocks down, and a first-measured candidate would look slow), then the candidates are timed in interleaved rounds."""
    live = {}
    C = torch.empty(nb * M * N, device=dev)
    skmax = max(c[1] for c in cands)
    ws = torch.empty(skmax * nb * M * N, device=dev) if skmax > 1 else None       # one output and one slab workspace shared by the candidates
    for c in cands:
        args, kw = setup(M, N, K, nb, akc, bkc, c[1], C, ws)
        try:
            L.gemm(*args, tile=c[0], **kw)
        except RuntimeError:
            continue
        live[c] = (args, kw)
    torch.cuda.synchronize()
    first = next(iter(live))
    t = burst(*live[first], first[0], 2)
    burst(*live[first], first[0], max(2, min(2000, int(30.0 / max(t, 1e-3)))))
    times = {c: [] for c in live}
    for _ in range(rounds):
        for c, (args, kw) in live.items():
            reps = 4 if not times[c] else max(4, min(48, int(0.6 / max(times[c][0], 1e-3))))
            times[c].append(burst(args, kw, c[0], reps))
    return {c: statistics.median(ts) for c, ts in times.items()}


for key, ms_total in sorted(shapes.items(), key=lambda kv: -kv[1]):
    M, N, K, nb, akc, bkc = key
    if 4.0 * nb * (M * K + N * K + 3 * M * N) > 24e9:
        continue
    # the planner's own pick
    import ctypes
    d = segx.GemmDesc(); d.M, d.N, d.K, d.nb0, d.nb1 = M, N, K, 1, nb
    d.a_b0, d.a_b1, d.a_m, d.a_k = (0, M * K, K, 1) if akc else (0, M * K, 1, M)
    d.b_b0, d.b_b1, d.b_n, d.b_k = (0, N * K, K, 1) if bkc else (0, N * K, 1, N)
    d.c_b0, d.c_b1, d.c_m = 0, M * N, N; d.alpha = 1.0
    t0, s0 = ctypes.c_int(0), ctypes.c_int(0)
    A0 = buf(nb * M * K); B0 = buf(nb * N * K)
    L.c.segx_gemm_plan_model(ctypes.c_void_p(A0.data_ptr()), ctypes.c_void_p(B0.data_ptr()), ctypes.byref(d), ctypes.byref(t0), ctypes.byref(s0))
    t1, s1 = ctypes.c_int(0), ctypes.c_int(0)                 # what the library does today (table of the previous sweep first)
    L.c.segx_gemm_plan(ctypes.c_void_p(A0.data_ptr()), ctypes.c_void_p(B0.data_ptr()), ctypes.byref(d), ctypes.byref(t1), ctypes.byref(s1))
    cur = (t1.value, s1.value)
    sks = sorted({1, s0.value, s1.value} | {s for s in (2, 3, 4, 6, 8, 12, 16, 24, 32) if K // s >= 128 and K >= 512})
    plan = (t0.value, s0.value)
    cands = [plan] + ([cur] if cur != plan else []) + [(tile, sk) for tile in (1, 5, 2, 6, 7, 8, 9) + ((10,) if akc else ()) + ((11,) if bkc else ()) if not (tile >= 6 and K % 32) for sk in sks
                      if not (sk > 1 and 4.0 * sk * nb * M * N > 6e9) and (tile, sk) != plan and (tile, sk) != cur]
    res = sweep(M, N, K, nb, akc, bkc, cands)
    if plan not in res:
        continue
    base = res[plan]
    bc = min(res, key=res.get)
    best = (res[bc], bc[0], bc[1])
    fl = 2.0 * M * N * K * nb
    print('shape %d %d %d %d %d %d  plan tile %d sk %d %.4f ms %.1f TF  best tile %d sk %d %.4f ms %.1f TF  gain %.3f  cur tile %d sk %d %.4f ms  weight %.2f' % (
        M, N, K, nb, int(akc), int(bkc), t0.value, s0.value, base, fl / base / 1e9, best[1], best[2], best[0], fl / best[0] / 1e9, base / best[0],
        cur[0], cur[1], res.get(cur, float('nan')), ms_total), flush=True)
