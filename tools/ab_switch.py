"""A/B of one operation-order switch on the SAME box and process (box-to-box clock differences are +-2.5 %):
python tools/ab_switch.py cfg4 InceptionModule.fuse_reductions [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import engine, functional as SF
from segtran_amd.efficientnet.model import MBConvBlock
from segtran_amd.networks.aj_i3d.aj_i3d import InceptionModule
from segtran_amd.networks import segtran_shared as ss

cfg, switch = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
owner, attr = switch.split('.')
owner = {'InceptionModule': InceptionModule, 'MBConvBlock': MBConvBlock, 'CrossAttFeatTrans': ss.CrossAttFeatTrans, '_ModesAggr': SF._ModesAggr, '_ConvStem2d': SF._ConvStem2d, '_BGemm': SF._BGemm, '_DWConv': SF._DWConv, '_PreNorm': SF._PreNorm, 'SF': SF}[owner]          # SF.block_nodes: one autograd node per backbone block
dev = torch.device('cuda', 0)
c = engine.CONFIGS[cfg]
batch = int(os.environ.get('SEGX_AB_BATCH', c['bs']))
torch.manual_seed(1); SF.manual_seed(1)
net = engine.build_model(cfg, dev); net.train()
step = engine.TrainStep(net, engine.init_optimizer(net, c['task']), c['task'])
x, raw = engine.synth_batch(cfg, batch, dev)
res = {}
for rep in range(int(os.environ.get('SEGX_AB_REPS', 2))):       # host-bound configurations (cfg1, one image per rank) need many short repetitions: read the minima
    for val in (True, False):
        setattr(owner, attr, val)
        for _ in range(4):
            step(x, raw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step(x, raw)
        torch.cuda.synchronize()
        res.setdefault(val, []).append((time.perf_counter() - t0) / steps * 1e3)
print('min on %.2f off %.2f' % (min(res[True]), min(res[False])))
print(cfg, 'batch', batch, switch, 'on: %s ms/step   off: %s ms/step' % (['%.2f' % v for v in res[True]], ['%.2f' % v for v in res[False]]))
