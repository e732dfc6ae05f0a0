"""ATen ops dispatched during one eager train step, grouped by the segtran_amd frame that issued them (TorchDispatchMode; backward on the calling thread).
GPU box: python tools/dispatch_count.py cfg2"""
import os, sys, collections, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import segx, engine, functional as SF
from torch.utils._python_dispatch import TorchDispatchMode
L = segx.lib(); L.set_engine('x6')
dev = torch.device('cuda', 0)
torch.autograd.set_multithreading_enabled(False)
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg = cfgname
torch.manual_seed(0); SF.manual_seed(0)
task = engine.CONFIGS[cfgname]['task']
net = engine.build_model(cfg, dev); net.train()
opt = engine.init_optimizer(net, task)
step = engine.TrainStep(net, opt, task)
x, raw = engine.synth_batch(cfg, engine.CONFIGS[cfgname]['bs'], dev)
step(x, raw); step(x, raw)
cnt = collections.Counter()
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.transpose', 'aten.as_strided', 'aten.slice', 'aten.select', 'aten.expand', 'aten.permute', 'aten.unsqueeze', 'aten.squeeze', 'aten.alias', 'aten.reshape', 'aten.empty', 'aten.split', 'aten.narrow', 'aten._local_scalar', 'aten.unbind', 'aten.is_', 'aten.sym_', 'aten.stride', 'aten.size', 'aten.new_empty', 'aten.empty_like', 'aten.lift_fresh', 'aten.unfold', 'aten.chunk')
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if 'segtran_amd' in f.filename]
            where = '%s:%d %s' % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].name) if fr else '(autograd engine)'
            big = any(isinstance(a, torch.Tensor) and a.numel() > 64 for a in list(args) + list((kwargs or {}).values()))
            cnt[(name, where, big)] += 1
        return func(*args, **(kwargs or {}))
with M():
    step(x, raw)
tot = sum(cnt.values())
print('aten ops dispatched in one step (views / empties excluded):', tot)
for (name, where, big), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print('%4d %-34s %-5s %s' % (n, name, 'big' if big else 'small', where))
