"""Device check of the EXPERIMENTAL bf16x6 GEMM entry point against the default fp32-MFMA path (tools only; not a test)."""
import sys, time, torch
sys.path.insert(0, '.')
from segtran_amd import segx
L = segx.lib()
dev = torch.device('cuda', 0)
g = torch.Generator(device='cpu').manual_seed(1)
for (M, N, K) in ((24576, 1792, 1792), (4096, 1792, 256), (1000, 300, 520)):
    A = torch.randn(M, K, generator=g).to(dev); B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    C0 = torch.empty(M, N, device=dev); C1 = torch.empty(M, N, device=dev)
    args = (M, N, K, (0, 0, K, 1), (0, 0, K, 1), (0, 0, N))
    L.use_bf16x6 = False; L.gemm(A, B, C0, *args)
    L.use_bf16x6 = True; L.gemm(A, B, C1, *args)
    torch.cuda.synchronize(); C1a = C1.clone()
    torch.cuda.synchronize()
    ts = []
    for flag in (False, True, 2):                       # fp32-MFMA, bf16x6 tile 128x128x32, bf16x6 tile 128x256x16
        L.c.segx_tune(3, 2 if flag == 2 else 1)
        L.use_bf16x6 = bool(flag)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.gemm(A, B, C1 if flag else C0, *args)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    L.use_bf16x6 = False; L.c.segx_tune(3, 1)
    ref = (A[:64].double() @ B.double().t())
    e0 = (C0[:64].double() - ref).abs().max().item() / ref.abs().max().item(); e1 = max((C1a[:64].double() - ref).abs().max().item(), (C1[:64].double() - ref).abs().max().item()) / ref.abs().max().item()
    fl = 2.0 * M * N * K
    print('%6d x %5d x %5d: fp32-MFMA %.3f ms (%.1f TF, err %.2e) | bf16x6 incl. split %.3f ms (%.1f TF, err %.2e) | wide tile %.3f ms (%.1f TF, err of last run %.2e)'
          % (M, N, K, ts[0], fl / ts[0] / 1e9, e0, ts[1], fl / ts[1] / 1e9, e1, ts[2], fl / ts[2] / 1e9, e1))
