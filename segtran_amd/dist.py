"""Data parallelism for the segtran train step: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

Reference behaviour being replaced: torch DDP + SyncBatchNorm (train2d.py:796-801, 1108-1113).  Here:
  * identical initial weights on every rank (same seed / synthetic weights), disjoint samples per rank;
  * gradients already live in ONE flat fp32 buffer (optimization.BertAdam.flat_grad); it is all-reduced in a few
    large buckets (xGMI links are point-to-point, ~153 GB/s each: few big collectives beat many small ones) and
    averaged; parameters that never receive gradients (N3) are simply zeros in the buffer -- no
    `find_unused_parameters` graph walk is needed;
  * BatchNorm statistics are synchronised inside libsegx's fused BN(+activation) op: one small all-gather per BN
    layer forward, one small all-reduce backward (`enable_sync_batchnorm`).
"""
import os
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """env:// rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), as torch.distributed.run provides."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 0, 1
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method='env://')
    return rank, local, world


class GradReducer:
    """Bucketed all-reduce (sum -> mean) of the optimizer's flat gradient buffer."""

    def __init__(self, optimizer, bucket_mb=128, group=None):
        self.flat = optimizer.flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = self.flat.numel()
        step = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self.buckets = [(o, min(n, o + step)) for o in range(0, n, step)]

    def allreduce_grads(self):
        if self.world <= 1:
            return
        works = [dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a, b in self.buckets]
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / self.world)


def reduce_scalars(t, group=None):
    """C3: ONE small all-reduce for all logged loss scalars (the reference issues one per scalar, train2d.py:1328-1337)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t


def enable_sync_batchnorm(group=None):
    """Synchronised BatchNorm for libsegx's fused BN(+act) op (replaces nn.SyncBatchNorm, train2d.py:1109).

    forward : ONE all-gather of [2C+1] floats per BN layer (mean, biased var, count) merged with Chan's formula;
    backward: ONE all-reduce of [2C] floats (sum du*xhat, sum du); the apply kernel then uses the global sums / count.
    Parameter gradients stay LOCAL sums (the flat-gradient all-reduce averages them like every other gradient)."""
    from . import functional as SF
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        SF._bn_stats_sync = SF._bn_grad_sync = None
        return False
    world = dist.get_world_size(group)

    def stats_sync(mean, var, n_local):
        C = mean.numel()
        loc = torch.cat([mean, var, mean.new_full((1,), float(n_local))])
        allv = loc.new_empty(world * (2 * C + 1))
        dist.all_gather_into_tensor(allv, loc, group=group)
        allv = allv.view(world, 2 * C + 1)
        means, vars_, ns = allv[:, :C], allv[:, C:2 * C], allv[:, 2 * C:]
        N = ns.sum()
        gmean = (means * ns).sum(0) / N
        gvar = ((vars_ + (means - gmean) ** 2) * ns).sum(0) / N
        # every rank holds the same per-GPU batch (bs // world, train2d.py:791): the global count is known on the
        # host without reading N back (a .item() here would be 96 host syncs per step)
        return gmean.contiguous(), gvar.contiguous(), int(n_local) * world

    def grad_sync(dw, db):
        both = torch.cat([dw, db])
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        C = dw.numel()
        return both[:C].contiguous(), both[C:].contiguous()

    SF._bn_stats_sync, SF._bn_grad_sync = stats_sync, grad_sync
    return True


def disable_sync_batchnorm():
    from . import functional as SF
    SF._bn_stats_sync = SF._bn_grad_sync = None
