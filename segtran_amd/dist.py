"""Data parallelism for the segtran train step: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

Reference behaviour being replaced: torch DDP + SyncBatchNorm (train2d.py:796-801, 1108-1113).  Here:
  * identical initial weights on every rank (same seed / synthetic weights), disjoint samples per rank;
  * gradients already live in ONE flat fp32 buffer (optimization.BertAdam.flat_grad); it is all-reduced in a few
    large buckets (xGMI links are point-to-point, ~153 GB/s each: few big collectives beat many small ones) and
    averaged; parameters that never receive gradients (N3) are simply zeros in the buffer -- no
    `find_unused_parameters` graph walk is needed;
  * BatchNorm statistics are synchronised with torch's SyncBatchNorm (plumbing; batching its per-layer
    collectives is listed in DESIGN.md as open work).
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


def convert_sync_batchnorm(net):
    if dist.is_initialized() and dist.get_world_size() > 1 and torch.cuda.is_available():
        return torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    return net
