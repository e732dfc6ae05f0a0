"""One-rank RCCL session (run by tests/test_gpu_dist.py in its own process): the 'nccl' (= RCCL) branch of segtran_amd/dist.py --
ReduceOp.AVG all-reduce of the gradient buckets launched from autograd hooks, all_gather_into_tensor / all_reduce of the synchronised
BatchNorm statistics on device tensors -- executed on the one GPU of the box with a world of size 1 (every collective is then the
identity), against the same steps without any collective.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segtran_amd import engine, dist as sdist, functional as SF      # noqa: E402


def losses(distributed, dev, cfg):
    torch.manual_seed(3); SF.manual_seed(3)
    net = engine.build_model(cfg, dev, dropout_prob=0.2, attractors=32)
    net.train()
    opt = engine.init_optimizer(net, cfg['task'], t_total=100, warmup_steps=2)
    reducer = None
    if distributed:
        assert sdist.enable_sync_batchnorm(force=True)
        reducer = sdist.GradReducer(opt, bucket_mb=8, force_collectives=True)
    step = engine.TrainStep(net, opt, cfg['task'], reducer)
    x, raw = engine.synth_batch(cfg, 2, dev)
    out = [float(step(x, raw).detach()) for _ in range(4)]
    info = dict(buckets=len(reducer.buckets), launched_in_backward=reducer._last_in_backward, avg=reducer._avg) if reducer else {}
    sdist.disable_sync_batchnorm()
    return out, info


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)
    res = {'backend': dist.get_backend()}
    for name, cfg in (('2d', dict(engine.CONFIGS['cfg2'], size=(64, 64))), ('3d', dict(engine.CONFIGS['cfg4'], size=(112, 112, 16)))):
        plain, _ = losses(False, dev, cfg)
        dp, info = losses(True, dev, cfg)
        res[name] = dict(plain=plain, rccl=dp, **info)
    t = sdist.reduce_scalars(torch.ones(3, device=dev))
    res['scalars'] = t.tolist()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == '__main__':
    main()
