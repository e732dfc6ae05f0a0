"""BertAdam on libsegx -- mirror of /root/reference/code/optimization.py (BertAdam :40-164, warmup_linear :25-31).

Same constructor arguments and param-group format as the reference class, same update rule (per-tensor
gradient clip `max_grad_norm`=0.05, Adam moments WITHOUT bias correction, decoupled weight decay,
warmup-linear LR, parameters without a gradient skipped entirely -- N3).  MI355X-native differences:
  * ONE multi-tensor step (3 kernel launches for all ~530 tensors) instead of ~10 ATen launches per tensor;
  * the trainer's preceding `nn.utils.clip_grad_norm_(net.parameters(), grad_clip)` (train2d.py:1324-1325) is
    folded into the same pass: pass `global_grad_clip=` to `step()` / the constructor;
  * data parallel: gradients live in ONE flat fp32 buffer (each `p.grad` is a view), which is what the reducer
    all-reduces in buckets over RCCL/xGMI (segtran_amd/dist.py);
  * single process (`release_flat_grads()`): autograd keeps ownership of every gradient tensor (`p.grad = None`
    before backward, so AccumulateGrad adopts the incoming tensor instead of issuing one `grad += g` kernel per
    parameter -- 549 launches / 3.3 ms per cfg2 step) and the multi-tensor step reads them through a pointer
    table refreshed each step;
  * data parallel, gathered (`use_gathered_grads()`, what `dist.GradReducer` selects): autograd owns the gradient
    tensors as above and the reducer copies each bucket's gradients into the flat buffer with ONE multi-tensor
    launch right before that bucket's all-reduce; the step then reads the (averaged) flat buffer.
"""
import torch
from torch.optim import Optimizer

from . import segx

CHUNK = 65536


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


def warmup_constant(x, warmup=0.002):
    return x / warmup if x < warmup else 1.0


SCHEDULES = {'warmup_linear': warmup_linear, 'warmup_constant': warmup_constant}


class BertAdam(Optimizer):
    def __init__(self, params, lr, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.05, max_grad_norm=0.05, global_grad_clip=0.0):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        self.global_grad_clip = global_grad_clip
        self.step_count = 0
        self._tabs = None
        self._touched = set()
        self._hooks = []
        self._private = False
        self._gathered = False
        self._ring, self._ring_pos = None, 0
        self._build_flat_grads()

    # ---- flat gradient buffer ---------------------------------------------------------------------------
    def _all_params(self):
        seen, out = set(), []
        for g in self.param_groups:
            for p in g['params']:
                if id(p) not in seen:
                    seen.add(id(p)); out.append((p, g))
        return out

    def _build_flat_grads(self):
        ps = self._all_params()
        dev = ps[0][0].device
        total = sum((p.numel() + 3) // 4 * 4 for p, _ in ps)             # 16-B aligned slices
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slices = []
        off = 0
        for p, _ in ps:
            n = p.numel()
            p.grad = self.flat_grad[off:off + n].view_as(p)
            self.slices.append((off, n))
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda q, s=self: s._touched.add(id(q))))
            off += (n + 3) // 4 * 4

    def use_gathered_grads(self):
        """Data-parallel mode of `dist.GradReducer(gather=True)`: p.grad are autograd's own tensors, the flat buffer is filled by
        the reducer's per-bucket gather and read by the step."""
        assert self._tabs is None and not self._private, 'use_gathered_grads() must precede the first step'
        self._gathered = True
        for p, _ in self._all_params():
            p.grad = None

    def _ensure_tables(self):
        if self._tabs is None:
            self._build_tables()
            for h in self._hooks:
                h.remove()                          # the touched set is static after the first backward (N3)
            self._hooks = []

    def release_flat_grads(self):
        """Single-process mode (no reducer): drop the flat views, autograd owns the gradient tensors from now on."""
        assert self._tabs is None or self._private, 'release_flat_grads() must precede the first step'
        self._private = True
        self.flat_grad = None
        for p, _ in self._all_params():
            p.grad = None

    def zero_grad(self, set_to_none=False):
        """Flat mode: gradients are views of one buffer, zero it (one memset) instead of dropping the tensors.
        Private mode: drop the tensors, so that the next backward adopts the fresh ones without an accumulate kernel."""
        if self._private or self._gathered:
            for p in (self._ps if self._tabs is not None else [p for p, _ in self._all_params()]):      # the cached list once the tables exist
                p.grad = None
        else:
            self.flat_grad.zero_()

    def _refresh_grad_table(self):
        """Private mode: this step's gradient addresses -> the device pointer table (async copy from a ring of pinned buffers)."""
        ps = self._ps
        f32 = torch.float32
        try:                        # one pass, no per-parameter branches on the good path (this loop is host time of EVERY eager step: ~600 parameters)
            gs = [p.grad if a else None for p, a in zip(ps, self._active)]
            ptrs = [0 if g is None else g.data_ptr() for g in gs]
            ok = all(g is None or (g.dtype is f32 and g.is_contiguous()) for g in gs)
        except AttributeError:
            ok = False
        if not ok or any(a and g is None for g, a in zip(gs, self._active)):
            for i, p in enumerate(ps):
                if self._active[i]:
                    if p.grad is None:
                        raise RuntimeError('parameter #%d received a gradient in the first step but none now: the set of trained '
                                           'parameters must stay fixed (N3)' % i)
                    assert p.grad.is_contiguous() and p.grad.dtype == torch.float32
        tab = self._tabs['grads']
        if tab.device.type != 'cuda':
            tab.copy_(torch.tensor(ptrs, dtype=torch.int64, device='cpu'))
            return
        if torch.cuda.is_current_stream_capturing():
            # hipGraph capture (engine.GraphedTrainStep): the gradient tensors of a captured step live at fixed addresses of the graph's private
            # pool, so the table written here is the table of every replay; no event bookkeeping (events cannot be waited on while capturing)
            self._graph_ptrs.copy_(torch.tensor(ptrs, dtype=torch.int64, device='cpu'))        # pinned buffer allocated by enter_graph_mode()
            tab.copy_(self._graph_ptrs, non_blocking=True)
            return
        if self._ring is None:
            self._ring = [(torch.empty(len(ps), dtype=torch.int64, device='cpu').pin_memory(), torch.cuda.Event()) for _ in range(4)]
        buf, ev = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        ev.synchronize()                                  # the copy issued 4 steps ago from this slot (no-op in practice)
        buf.copy_(torch.tensor(ptrs, dtype=torch.int64, device='cpu'))
        tab.copy_(buf, non_blocking=True)
        ev.record()

    def _build_tables(self):
        ps = self._all_params()
        dev = self.flat_m.device
        i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)      # noqa: E731
        i32 = lambda xs: torch.tensor(xs, dtype=torch.int32, device=dev)      # noqa: E731
        f32 = lambda xs: torch.tensor(xs, dtype=torch.float32, device=dev)    # noqa: E731
        mp, vp = self.flat_m.data_ptr(), self.flat_v.data_ptr()
        gp = 0 if self._private else self.flat_grad.data_ptr()
        self._ps = [p for p, _ in ps]
        self._active = [id(p) in self._touched for p, _ in ps]
        chunk_tensor, chunk_off, chunk_first = [], [], [0]
        for t, ((p, g), (off, n)) in enumerate(zip(ps, self.slices)):
            assert p.is_contiguous() and p.dtype == torch.float32
            assert self._private or self._gathered or (p.grad is not None and p.grad.data_ptr() == gp + 4 * off), 'p.grad was re-bound; use optimizer.zero_grad()'
            for c in range(0, n, CHUNK):
                chunk_tensor.append(t); chunk_off.append(c)
            chunk_first.append(len(chunk_tensor))
        self._tabs = dict(
            params=i64([p.data_ptr() for p, _ in ps]), grads=i64([gp + 4 * o for o, _ in self.slices]),
            m=i64([mp + 4 * o for o, _ in self.slices]), v=i64([vp + 4 * o for o, _ in self.slices]),
            sizes=i64([n for _, n in self.slices]), chunk_tensor=i32(chunk_tensor), chunk_off=i64(chunk_off),
            chunk_first=i32(chunk_first), active=i32([1 if id(p) in self._touched else 0 for p, _ in ps]),
            lr=f32([g['lr'] for _, g in ps]), wd=f32([g['weight_decay'] for _, g in ps]))
        self._lrwd_host = self._lrwd_key()
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:          # one schedule / one set of Adam constants per step launch
            for k in ('schedule', 'warmup', 't_total', 'b1', 'b2', 'e', 'max_grad_norm'):
                if g[k] != g0[k]:
                    raise ValueError("BertAdam (libsegx multi-tensor step): param groups must share '%s' (%r vs %r); only lr and "
                                     "weight_decay may differ between groups" % (k, g[k], g0[k]))
        self._nt, self._nch = len(ps), len(chunk_tensor)
        self._chunk_first_host = chunk_first
        self._ws = torch.zeros(self._nch + 2 * self._nt + 2, dtype=torch.float32, device=dev)
        self._active_names = None

    def _lrwd_key(self):
        return tuple((float(g['lr']), float(g['weight_decay'])) for g in self.param_groups)

    def _sync_lr_wd(self):
        """param_groups[i]['lr' | 'weight_decay'] edited after the first step (LR schedulers, manual decay): refresh the per-tensor device
        tables the kernel reads, so that get_lr() and the update cannot diverge."""
        key = self._lrwd_key()
        if key != self._lrwd_host:
            ps = self._all_params()
            dev = self.flat_m.device
            self._tabs['lr'].copy_(torch.tensor([g['lr'] for _, g in ps], dtype=torch.float32, device='cpu').to(dev))
            self._tabs['wd'].copy_(torch.tensor([g['weight_decay'] for _, g in ps], dtype=torch.float32, device='cpu').to(dev))
            self._lrwd_host = key

    # ---- checkpointing: the reference's per-parameter layout (state[p] = {'step', 'next_m', 'next_v'}, optimization.py:108-114) ------
    def state_dict(self):
        """Same wire format as the reference BertAdam (torch.optim state_dict with 'step' / 'next_m' / 'next_v' per parameter), built from
        the flat moment buffers; parameters that never received a gradient carry no state, as in the reference (N3)."""
        ps = self._all_params()
        if self._tabs is not None:
            active = self._active
        else:                                            # loaded but not stepped yet: a parameter is trained iff the loaded moments are non-zero (N3)
            active = [bool(self.flat_v[off:off + n].any()) if self.step_count > 0 else True for (off, n) in self.slices]
        self.state.clear()
        if self.step_count > 0:
            for (p, _), (off, n), a in zip(ps, self.slices, active):
                if a:
                    self.state[p] = dict(step=self.step_count, next_m=self.flat_m[off:off + n].view_as(p).clone(),
                                         next_v=self.flat_v[off:off + n].view_as(p).clone())
        sd = super().state_dict()
        self.state.clear()
        return sd

    def load_state_dict(self, state_dict):
        """Accepts what state_dict() writes and what the reference's BertAdam writes ('optim_state' of a reference checkpoint,
        train2d.py:629-635): moments go into the flat buffers, the step count (= position on the warm-up / decay schedule) is restored."""
        super().load_state_dict(state_dict)
        ps = self._all_params()
        steps = []
        with torch.no_grad():
            for (p, _), (off, n) in zip(ps, self.slices):
                st = self.state.get(p)
                if not st:
                    continue
                self.flat_m[off:off + n].copy_(st['next_m'].reshape(-1))
                self.flat_v[off:off + n].copy_(st['next_v'].reshape(-1))
                steps.append(int(st['step']))
        self.state.clear()
        if steps:
            if min(steps) != max(steps):
                raise ValueError('BertAdam.load_state_dict: parameters carry different step counts (%d..%d); the multi-tensor step keeps one '
                                 'schedule position' % (min(steps), max(steps)))
            self.step_count = steps[0]
        if self._tabs is not None:
            self._lrwd_host = None                         # param_groups were replaced: refresh the device tables at the next step

    def get_lr(self):
        g = self.param_groups[0]
        if self.step_count == 0:
            return [0]
        sched = SCHEDULES[g['schedule']](self.step_count / g['t_total'], g['warmup']) if g['t_total'] != -1 else 1.0
        return [gr['lr'] * sched for gr in self.param_groups]

    def grad_norm(self):
        """Global gradient norm of the last step (device tensor; reading it synchronises)."""
        return self._ws[self._nch + 2 * self._nt]

    # ---- hipGraph mode: the schedule factor cannot be a kernel argument of a replayed launch -> it is folded into the lr table ----------
    def enter_graph_mode(self):
        """After this call step() launches with schedule factor 1 and reads lr[t] * schedule(step) from the device table that
        prepare_replay() refreshes (one 2-KB async copy on the step's stream) before every replay of the captured step."""
        self._ensure_tables()
        self._graph_mode = True
        ps = self._all_params()
        self._lr_host = torch.tensor([g['lr'] for _, g in ps], dtype=torch.float32, device='cpu')
        self._lr_pinned = [torch.empty_like(self._lr_host).pin_memory() for _ in range(4)] if self.flat_m.is_cuda else None
        self._lr_events = [None] * 4                    # per pinned slot: the event recorded after its H2D copy (the host may run replays ahead)
        self._lr_slot = 0
        self._graph_ptrs = torch.empty(len(ps), dtype=torch.int64, device='cpu')
        if self.flat_m.is_cuda:
            self._graph_ptrs = self._graph_ptrs.pin_memory()
        self.prepare_replay(advance=False)

    def prepare_replay(self, advance=True):
        g = self.param_groups[0]
        self._refresh_graph_lr_wd()
        sched = SCHEDULES[g['schedule']](self.step_count / g['t_total'], g['warmup']) if g['t_total'] != -1 else 1.0
        if self._lr_pinned is None:
            self._tabs['lr'].copy_(self._lr_host * sched)
        else:
            # a pinned slot is rewritten only after the copy that last read it has executed: replays are enqueued far faster than the device
            # runs them (bench.py --graph loops without synchronising), and an unguarded ring of four would hand later steps' rates to earlier ones
            i = self._lr_slot; self._lr_slot = (i + 1) % len(self._lr_pinned)
            if self._lr_events[i] is not None:
                self._lr_events[i].synchronize()
            buf = self._lr_pinned[i]
            torch.mul(self._lr_host, sched, out=buf)
            self._tabs['lr'].copy_(buf, non_blocking=True)
            if self._lr_events[i] is None:
                self._lr_events[i] = torch.cuda.Event()
            self._lr_events[i].record()
        if advance:
            self.step_count += 1

    def _refresh_graph_lr_wd(self):
        """Graph mode: param_group edits (lr / weight_decay, load_state_dict) reach the device tables here -- step() no longer runs _sync_lr_wd."""
        ps = self._all_params()
        key = tuple((g['lr'], g['weight_decay']) for _, g in ps)
        if key != getattr(self, '_graph_lrwd_key', None):
            self._lr_host = torch.tensor([g['lr'] for _, g in ps], dtype=torch.float32, device='cpu')
            self._tabs['wd'].copy_(torch.tensor([g['weight_decay'] for _, g in ps], dtype=torch.float32, device='cpu').to(self._tabs['wd'].device))
            self._graph_lrwd_key = key

    @torch.no_grad()
    def step(self, closure=None, global_grad_clip=None):
        loss = closure() if closure is not None else None
        segx.lib().team_check()           # a team BatchNorm exchange of an earlier launch timed out -> RuntimeError here, not NaN weights (one host word, no sync)
        self._ensure_tables()
        if self._private:
            self._refresh_grad_table()
        g = self.param_groups[0]
        if getattr(self, '_graph_mode', False):
            clip = self.global_grad_clip if global_grad_clip is None else global_grad_clip
            segx.lib().mt_bertadam_step(self._tabs, self._nt, self._nch, CHUNK, float(clip), float(g['max_grad_norm']),
                                        1.0, g['b1'], g['b2'], g['e'], self._ws)
            return loss                                   # step_count advances in prepare_replay()
        self._sync_lr_wd()
        sched = SCHEDULES[g['schedule']](self.step_count / g['t_total'], g['warmup']) if g['t_total'] != -1 else 1.0
        clip = self.global_grad_clip if global_grad_clip is None else global_grad_clip
        segx.lib().mt_bertadam_step(self._tabs, self._nt, self._nch, CHUNK, float(clip), float(g['max_grad_norm']),
                                    float(sched), g['b1'], g['b2'], g['e'], self._ws)
        self.step_count += 1
        return loss
