"""Where the ATen launches of a train step come from: torch.profiler over one eager step of a bench configuration, aten ops that launch device work grouped by the
Python frame that called them.  GPU box:  python tools/host_profile.py cfg2 [batch]"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import engine, segx, functional as SF
from torch.profiler import profile, ProfilerActivity

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else engine.CONFIGS[cfg]['bs']
dev = torch.device('cuda', 0)
segx.lib().set_engine('x6')
torch.manual_seed(0); SF.manual_seed(0)
net = engine.build_model(cfg, dev); net.train()
opt = engine.init_optimizer(net, engine.CONFIGS[cfg]['task'])
step = engine.TrainStep(net, opt, engine.CONFIGS[cfg]['task'])
x, raw = engine.synth_batch(cfg, B, dev)
for _ in range(3):
    step(x, raw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, raw)
    torch.cuda.synchronize()
by = collections.Counter(); tm = collections.Counter()
def dev_us(e):
    for a in ('self_device_time_total', 'self_cuda_time_total', 'device_time_total', 'cuda_time_total'):
        v = getattr(e, a, 0) or 0
        if v > 0:
            return v
    return sum(getattr(k, 'duration', 0) or 0 for k in (getattr(e, 'kernels', None) or []))


for e in prof.events():
    if not e.name.startswith('aten::') or e.cpu_children or dev_us(e) <= 0:
        continue
    frames = [f for f in (e.stack or []) if 'segtran_amd' in f or 'bench.py' in f]
    where = frames[0].strip() if frames else '(autograd engine / no python frame)'
    by[(e.name, where)] += 1; tm[(e.name, where)] += dev_us(e)
print('aten ops with device time, one %s step (batch %d): count, device us, op, caller' % (cfg, B))
for k, n in sorted(by.items(), key=lambda kv: -tm[kv[0]])[:60]:
    print('%4d %9.1f  %-28s %s' % (n, tm[k], k[0], k[1][:150]))
print('total aten device launches', sum(by.values()), 'device us', round(sum(tm.values()), 1))
