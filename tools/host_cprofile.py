"""Host time of the EAGER train step (the data-parallel step is eager: RCCL cannot be captured): cProfile over a few steps of a configuration whose
kernels are shorter than its Python (cfg1, cfg3 at one image per rank), functions ranked by their own time.  GPU box:
    python tools/host_cprofile.py cfg1 [batch] [steps]"""
import os, sys, time, cProfile, pstats, io, gc, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import engine, segx, functional as SF

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
B = int(sys.argv[2]) if len(sys.argv) > 2 else engine.CONFIGS[cfg]['bs']
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda', 0)
segx.lib().set_engine('x6')
SF.block_nodes = os.environ.get('SEGX_BLOCK_NODES', '1') != '0'          # A/B: one autograd node per backbone block vs one per op
torch.manual_seed(0); SF.manual_seed(0)
net = engine.build_model(cfg, dev); net.train()
opt = engine.init_optimizer(net, engine.CONFIGS[cfg]['task'])
step = engine.TrainStep(net, opt, engine.CONFIGS[cfg]['task'])
x, raw = engine.synth_batch(cfg, B, dev)
for _ in range(5):
    step(x, raw)
torch.cuda.synchronize()
gc.collect(); gc.disable()
t0 = time.perf_counter()
for _ in range(K):
    step(x, raw)
t_host = time.perf_counter() - t0                 # host time to ISSUE K steps (the queue is deep enough not to block)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('block_nodes', SF.block_nodes)
print('%s batch %d: %.2f ms/step wall, %.2f ms/step of host issue time' % (cfg, B, t_all / K * 1e3, t_host / K * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step(x, raw)
pr.disable()
torch.cuda.synchronize()
gc.enable()
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print('==== by %s (%d steps)' % (key, K))
    print(s.getvalue()[:9000])
