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
    # SEGX_DIST_BACKEND=gloo lets two ranks share ONE GPU (RCCL refuses duplicate devices): used by the single-GPU test of this path
    backend = backend or os.environ.get('SEGX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method='env://')
    return rank, local, world


class GradReducer:
    """Bucketed all-reduce (-> mean) of the optimizer's flat gradient buffer, OVERLAPPED with backward.

    Step 1 runs the buckets after backward and learns which parameters receive gradients at all (N3: the set is static).  From
    step 2 on every such parameter carries a post-accumulate hook; a bucket's all-reduce is launched (async, on RCCL's stream) the
    moment the last of its parameters has its gradient -- backward produces them in reverse registration order, so the head's
    buckets travel over xGMI while the backbone is still differentiating.  `allreduce_grads()` then only launches what is left and
    waits.  Buckets that hold no live parameter are never sent (they are zeros on every rank)."""

    def __init__(self, optimizer, bucket_mb=64, group=None, overlap=True):
        self.opt, self.flat, self.group, self.overlap = optimizer, optimizer.flat_grad, group, overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = self.flat.numel()
        step = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self.buckets = [(o, min(n, o + step)) for o in range(0, n, step)]
        self._avg = dist.is_initialized() and dist.get_backend(group) == 'nccl'        # RCCL averages in the collective; gloo has no AVG
        self._armed, self._hooks = False, []
        self._pending, self._need, self._works, self._launched = [], [], [], []
        self.launched_in_backward = 0                                                  # of the last step (tests / logging)

    def _launch(self, k):
        a, b = self.buckets[k]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append(dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=True))
        self._launched[k] = True

    def _arm(self):
        """after the first backward: count the live parameters of each bucket and hook them"""
        ps, touched = self.opt._all_params(), self.opt._touched
        self._need = [0] * len(self.buckets)
        for (p, _), (off, n) in zip(ps, self.opt.slices):
            if id(p) not in touched or n == 0:
                continue
            ks = [k for k, (a, b) in enumerate(self.buckets) if a < off + n and off < b]
            for k in ks:
                self._need[k] += 1
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda q, ks=ks: self._on_grad(ks)))
        self._pending = list(self._need)
        self._launched = [False] * len(self.buckets)
        self._armed = True

    def _on_grad(self, ks):
        for k in ks:
            self._pending[k] -= 1
            if self._pending[k] == 0 and not self._launched[k]:
                self._launch(k)
                self.launched_in_backward += 1

    def allreduce_grads(self):
        if self.world <= 1:
            return
        if not self._armed:
            self._launched = [False] * len(self.buckets)
            live = range(len(self.buckets))
        else:
            live = [k for k, need in enumerate(self._need) if need > 0]
        for k in live:
            if not self._launched[k]:
                self._launch(k)
        for w in self._works:
            w.wait()
        if not self._avg:
            self.flat.mul_(1.0 / self.world)
        self._works = []
        if not self._armed and self.overlap:
            self._arm()
        else:
            self._pending = list(self._need)
            self._launched = [False] * len(self.buckets)
        self._last_in_backward, self.launched_in_backward = self.launched_in_backward, 0


def reduce_scalars(t, group=None):
    """C3: ONE small all-reduce for all logged loss scalars (the reference issues one per scalar, train2d.py:1328-1337)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t


def enable_sync_batchnorm(group=None):
    """Synchronised BatchNorm for libsegx's fused BN(+act) op (replaces nn.SyncBatchNorm, train2d.py:1109).

    forward : ONE all-gather of [2C] floats per BN layer (mean, biased var), merged with Chan's formula by one kernel;
    backward: ONE all-reduce of [2C] floats (sum du*xhat, sum du); the apply kernel then uses the global sums / count.
    Parameter gradients stay LOCAL sums (the flat-gradient all-reduce averages them like every other gradient)."""
    from . import functional as SF
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        SF._bn_stats_sync = SF._bn_grad_sync = None
        return False
    world = dist.get_world_size(group)

    def stats_sync(loc):
        """loc [2C] = this rank's (mean, biased var) -> (all [world, 2C], world).  Every rank holds the same per-GPU batch
        (bs // world, train2d.py:791), so the counts need not travel and the merge (segx_bn_merge_stats) is one kernel."""
        allv = loc.new_empty(world * loc.numel())
        dist.all_gather_into_tensor(allv, loc, group=group)
        return allv, world

    def grad_sync(both):
        """both [2C] = local (sum du*xhat, sum du) -> global sums (a new tensor: the local sums remain the parameter gradients)."""
        glob = both.clone()
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=group)
        return glob

    SF._bn_stats_sync, SF._bn_grad_sync = stats_sync, grad_sync
    return True


def disable_sync_batchnorm():
    from . import functional as SF
    SF._bn_stats_sync = SF._bn_grad_sync = None
