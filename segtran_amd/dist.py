"""Data parallelism for the segtran train step: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

Reference behaviour being replaced: torch DDP + SyncBatchNorm (train2d.py:796-801, 1108-1113).  Here:
  * identical initial weights on every rank (same seed / synthetic weights), disjoint samples per rank;
  * gradients travel through ONE flat fp32 buffer (optimization.BertAdam.flat_grad) cut into a few large buckets of whole
    parameters (xGMI links are point-to-point, ~153 GB/s each: few big collectives beat many small ones); autograd owns the
    gradient tensors, a bucket is filled by ONE multi-tensor gather launch when its last gradient is ready, all-reduced
    (averaged) while backward continues, and the optimizer reads the buffer; parameters that never receive gradients (N3)
    are simply zeros in the buffer -- no `find_unused_parameters` graph walk is needed;
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

    def __init__(self, optimizer, bucket_mb=64, group=None, overlap=True, gather=True, force_collectives=False, overlap_bn_teams=True):
        """gather=True: gradients stay autograd's own tensors and are copied into their bucket by one multi-tensor launch per
        bucket (no per-parameter `grad += g` kernels); gather=False: every p.grad is a view of the flat buffer.
        force_collectives: issue the all-reduces even in a one-rank group (the single-GPU RCCL test: AVG over one rank is the identity)."""
        self.opt, self.flat, self.group, self.overlap, self.gather = optimizer, optimizer.flat_grad, group, overlap, gather
        self.force = bool(force_collectives) and dist.is_initialized()
        if gather:
            optimizer.use_gathered_grads()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # the host-side (gloo) twin of the group for check_equal_batch, created HERE: dist.new_group is collective over the DEFAULT group, and the
        # reducer is constructed by every rank at the same point of the program (a lazy creation inside the first check would hang a strict sub-group)
        self.host_group = _host_group(group) if dist.is_initialized() and self.world > 1 else None
        step = max(1, int(bucket_mb * 1024 * 1024 // 4))
        # buckets of WHOLE parameters (flat range a:b, parameter range i0:i1), closed once they reach bucket_mb
        self.buckets, a, i0 = [], 0, 0
        slices = optimizer.slices
        for i, (off, n) in enumerate(slices):
            end = off + (n + 3) // 4 * 4
            if end - a >= step or i == len(slices) - 1:
                self.buckets.append((a, end, i0, i + 1))
                a, i0 = end, i + 1
        self._avg = dist.is_initialized() and dist.get_backend(group) == 'nccl'        # RCCL averages in the collective; gloo has no AVG
        # Collectives launched from the hooks run RCCL's reduction kernels on the compute units WHILE backward's kernels run.  The one-launch TEAM form of training
        # BatchNorm (backbone.hip; only reached WITHOUT synchronised BatchNorm -- the synchronised form never uses teams) needs its workgroups co-resident, which a
        # concurrent kernel can delay but not prevent: a team is at most half the CUs' worth of workgroups at two resident workgroups per CU (team_cap), RCCL's
        # channels hold a few dozen, and the polls are bounded and fail loudly (segx_team_status -> RuntimeError at the optimizer step).  r06:
        # the teams therefore STAY on by default -- the per-rank step keeps the 1 + 1 / 2 + 1 pass structure of the one-GPU step; r05 switched them off whenever
        # overlap was on (3 ms per cfg2 step).  `overlap_bn_teams=False` restores that (two-launch form, knob 3 = 1) for a stack where the check does fail.
        self.bn_teams_off, self._bn_path_prev = False, 0
        if overlap and (self.world > 1 or self.force) and self.flat.is_cuda and not overlap_bn_teams:
            from . import segx
            c = segx.lib().c
            self._bn_path_prev = c.segx_tune_get(3)                 # whatever SEGX_TUNE / a bench loop / a test put there: restored by close(), left alone unless it is the default
            if self._bn_path_prev == 0:
                self.bn_teams_off = c.segx_tune(3, 1) == 0
        self._armed, self._hooks = False, []
        self._pending, self._need, self._works, self._launched = [], [], [], []
        self._src, self._slot = None, 0
        self.launched_in_backward = 0                                                  # of the last step (tests / logging)

    # ---- gather: this step's gradient addresses -> device table -> one multi-tensor copy per bucket -----------------------
    def _src_slot(self):
        dev = self.flat.device
        if self._src is None:
            nt = len(self.opt.slices)
            mk = lambda: torch.zeros(nt, dtype=torch.int64, device='cpu')          # noqa: E731
            self._src = [dict(dev=torch.zeros(nt, dtype=torch.int64, device=dev), host=mk().pin_memory() if dev.type == 'cuda' else mk(),
                              ev=torch.cuda.Event() if dev.type == 'cuda' else None, fresh=True) for _ in range(4)]
        slot = self._src[self._slot]
        if slot['fresh']:
            if slot['ev'] is not None:
                slot['ev'].synchronize()            # the copies issued from this slot four steps ago (a no-op in practice)
            slot['fresh'] = False
        return slot

    def _gather(self, k):
        from . import segx
        opt = self.opt
        opt._ensure_tables()
        a, b, i0, i1 = self.buckets[k]
        slot = self._src_slot()
        ptrs = []
        for p in opt._ps[i0:i1]:
            g = p.grad
            if g is not None and not (g.is_contiguous() and g.dtype == torch.float32):
                raise RuntimeError('gradient of a non-contiguous / non-fp32 layout cannot be gathered')
            ptrs.append(0 if g is None else g.data_ptr())
        slot['host'][i0:i1] = torch.tensor(ptrs, dtype=torch.int64, device='cpu')
        slot['dev'][i0:i1].copy_(slot['host'][i0:i1], non_blocking=True)
        if slot['ev'] is not None:
            slot['ev'].record()
        c0, c1 = opt._chunk_first_host[i0], opt._chunk_first_host[i1]
        if c1 > c0:
            from .optimization import CHUNK
            segx.lib().mt_gather(slot['dev'], opt._tabs, c0, c1 - c0, CHUNK)

    def _launch(self, k):
        a, b = self.buckets[k][:2]
        if self.gather:
            self._gather(k)
        if self.world > 1 or self.force:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._works.append(dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=True))
        self._launched[k] = True

    def _arm(self):
        """after the first backward: count the live parameters of each bucket and hook them"""
        ps, touched = self.opt._all_params(), self.opt._touched
        self._need = [0] * len(self.buckets)
        for k, (a, b, i0, i1) in enumerate(self.buckets):
            for (p, _), (off, n) in zip(ps[i0:i1], self.opt.slices[i0:i1]):
                if id(p) not in touched or n == 0:
                    continue
                self._need[k] += 1
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda q, k=k: self._on_grad(k)))
        self._pending = list(self._need)
        self._launched = [False] * len(self.buckets)
        self._armed = True

    def _on_grad(self, k):
        self._pending[k] -= 1
        if self._pending[k] == 0 and not self._launched[k]:
            self._launch(k)
            self.launched_in_backward += 1

    def allreduce_grads(self):
        if self.world <= 1 and not self.gather:
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
        if not self._avg and self.world > 1:
            self.flat.mul_(1.0 / self.world)
        self._works = []
        if self._src is not None:
            self._slot = (self._slot + 1) % len(self._src)
            self._src[self._slot]['fresh'] = True
        if not self._armed and self.overlap:
            self._arm()
        else:
            self._pending = list(self._need)
            self._launched = [False] * len(self.buckets)
        self._last_in_backward, self.launched_in_backward = self.launched_in_backward, 0

    def overlap_stats(self):
        """{'buckets': live buckets, 'launched_in_backward': of the last step, 'bucket_mb': their mean size} -- bench.py's overlap evidence"""
        live = [k for k, need in enumerate(self._need) if need > 0] if self._armed else list(range(len(self.buckets)))
        mb = sum(self.buckets[k][1] - self.buckets[k][0] for k in live) * 4 / 2 ** 20 / max(1, len(live))
        return {'buckets': len(live), 'launched_in_backward': getattr(self, '_last_in_backward', 0), 'bucket_mb': round(mb, 1), 'bn_teams_off': self.bn_teams_off}

    def close(self):
        """remove the hooks and give BatchNorm its default forms back"""
        for h in self._hooks:
            h.remove()
        self._hooks, self._armed = [], False
        if self.bn_teams_off:
            from . import segx
            segx.lib().c.segx_tune(3, self._bn_path_prev)
            self.bn_teams_off = False


def reduce_scalars(t, group=None):
    """C3: ONE small all-reduce for all logged loss scalars (the reference issues one per scalar, train2d.py:1328-1337)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t


_HOST_GROUPS = {}                  # group object (None = the default group) -> its gloo twin; the key keeps the group alive, so it cannot be recycled


def _host_group(group=None):
    """A gloo twin of `group` for tiny host-side agreements: its collectives never touch the GPU streams.  dist.new_group is collective over the
    whole DEFAULT group -- call this where every rank passes (GradReducer.__init__ does), not from inside a sub-group's step."""
    if dist.get_backend(group) == 'gloo':
        return group
    if group not in _HOST_GROUPS:
        ranks = list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group)
        _HOST_GROUPS[group] = dist.new_group(ranks=ranks, backend='gloo')
    return _HOST_GROUPS[group]


def check_equal_batch(n, group=None, host_group=None):
    """Every rank must hold the same per-GPU batch (train2d.py:791 `bs // world_size`): the gradient AVG all-reduce, the synchronised
    BatchNorm merge and its backward all weight the ranks equally.  Called by TrainStep on EVERY step and on EVERY rank -- collective-safe: a
    check entered only by the ranks whose batch changed (a ragged last batch) would be matched with the other ranks' next collective.  One
    2-float MAX all-reduce of (n, -n) on a host-side (gloo) group: no device synchronisation, ~0.1 ms of host time per step."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    t = torch.tensor([float(n), -float(n)], dtype=torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_group if host_group is not None else _host_group(group))
    hi, lo = t.tolist()
    if hi != -lo:
        raise RuntimeError('data-parallel step with unequal per-rank batch sizes (between %d and %d, this rank %d): gradients and synchronised '
                           'BatchNorm statistics would be mis-weighted (drop or pad the ragged batch, as DistributedSampler does)' % (-lo, hi, n))


def enable_sync_batchnorm(group=None, force=False):
    """Synchronised BatchNorm for libsegx's fused BN(+act) op (replaces nn.SyncBatchNorm, train2d.py:1109).

    forward : ONE all-gather of [C] float4 partials (n, mean, M2) per BN layer, merged with Chan's formula inside the apply pass (no merge launch);
    backward: ONE all-reduce of [2C] floats (sum du*xhat, sum du); the apply kernel then uses the global sums / count.
    Parameter gradients stay LOCAL sums (the flat-gradient all-reduce averages them like every other gradient)."""
    from . import functional as SF
    if not dist.is_initialized() or (dist.get_world_size(group) <= 1 and not force):      # force: the one-rank RCCL test
        SF._bn_stats_sync = SF._bn_grad_sync = None
        return False
    world = dist.get_world_size(group)

    def stats_sync(loc):
        """loc [C] float4 = this rank's (n, mean, M2) partial per channel (segx_bn_stats_local) -> (all [world][C] float4, world): the BatchNorm
        apply pass merges the ranks' partials itself (segx_bn_act_fwd2, nparts = -world); the counts travel with the partials."""
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
