#!/usr/bin/env python
"""bench.py -- train-step throughput of the MI355X-native Segtran hot path (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under torch.distributed.run (one rank per GPU over
RCCL); started WITHOUT a launcher (`WORLD_SIZE` unset) it spawns the N ranks itself (self_spawn).  A step is the full train step: forward -> BCE+Dice -> backward -> gradient all-reduce -> global clip + BertAdam, fp32 results,
train mode with the reference's dropout 0.2, synthetic inputs / name-hashed synthetic weights (no network), inputs resident in HBM
before the timed region.  Rank 0 prints ONE JSON line:

  * the MAIN measurement (`metric`, `value`, ...): BASELINE.json configs[1] -- REFUGE fundus 2D, eff-b4, --translayers 3 --layercompress
    1,1,2,2, 512 x 512, batch 6 PER GPU (weak scaling) -- W warm-up steps, then EXACTLY K steps inside one barrier + synchronize bracket;
    `value` = images of all ranks / that time (max over ranks); per-step HIP-event times give `ms_per_step_median` beside it;
  * `"brats"`: the 3-D half of the metric measured the same way in the same process (own `roofline`): cfg4 (BraTS 112 x 112 x 96, bs 4, one
    layer, BASELINE configs[3]) at N = 1, and cfg5 (128^3, 4 volumes per GPU, two layers, BASELINE configs[4]: the configuration the
    >= 6x scaling target is stated on) at every N;
  * `"polyp"` (N > 1 only): BASELINE configs[2], polyp 352 x 352, at 6 images per GPU and with the reference's global batch 6 split over the ranks;
  * `roofline`: the dominant kernel family = the tile engine (dense + implicit-GEMM kernels), with `attention_gemms`: up to six of the mode-batched
    squeezed-attention QK^T / P.V GEMMs as [M, N, K, batch, ms_per_step, TFLOP/s, frac]; `cpu_baseline`: the oracle on the host cores.
The printed line is the COMPACT form (compact_line(): < 4 KB, the driver keeps an ~8 KB tail); the whole result -- `by_shape` (the twelve engine shapes
with the most time, each with its FLOP per byte and applicable roof), notes, per-step extremes -- is written to bench_shapes.json beside this file.
`--config cfgN` makes another BASELINE configuration the main measurement; `--engine f32` times the fp32-MFMA engine instead of bf16x6.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROF_EVERY = 10                     # engine launches carry HIP events on every 10th timed step (measure()): two event records per launch cost ~1.7 ms of host time on such a step
PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_TBS = 8.0                  # same guide: HBM3E ~8 TB/s (TFLOP/s per FLOP/byte)
PEAK_BF16_MFMA_TFLOPS = 2500.0      # same table: bf16 MFMA dense (32x32x16); the bf16x6 engine issues 6 of them per fp32-equivalent block product
WORKLOADS = {'cfg1': 'REFUGE fundus 2D, segtran eff-b4, translayers 1, 256x256, bs 2/GPU',
             'cfg2': 'REFUGE fundus 2D, segtran eff-b4, translayers 3, layercompress 1,1,2,2, 512x512, bs 6/GPU',
             'cfg3': 'Polyp 2D, segtran eff-b4, translayers 3, layercompress 1,1,2,2, 352x352, bs 6/GPU',
             'cfg4': 'BraTS 3D, segtran i3d, translayers 1, attractors 1024, 112x112x96 x4 modalities, bs 4/GPU',
             'cfg5': 'BraTS 3D, segtran i3d, translayers 2, attractors 1024, 128x128x128 x4 modalities, bs 4/GPU'}


def _cpu_steps(cfg_name, batch, threads, n_timed, budget_s):
    """oracle/ (CPU restatement of the reference, kind='port'): full train steps (fwd + loss + bwd + global clip + BertAdam) at `batch`
    samples of cfg_name's shapes / synthetic weights on the host cores: ONE untimed warm-up step (the first iterations of the reference's own
    CPU path run 2-3x slower while the allocators warm up), then up to n_timed timed steps within budget_s seconds."""
    from oracle import segtran_oracle as O
    from segtran_amd import engine
    from segtran_amd.synth import synth_state_dict
    c = engine.CONFIGS[cfg_name]
    torch.set_num_threads(threads)
    net = engine.build_model(cfg_name, 'cpu', synth=False)          # only for the key/shape list (no kernels run)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    del net
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    x, raw = engine.synth_batch(cfg_name, batch, 'cpu')
    mask = engine.map_mask(c['task'], raw)
    pw, _ = engine.loss_weights(c['task'], 'cpu')
    dims = [1792 if c['dim'] == 2 else 1024]
    for r in c['compress'][1:]:
        dims.append(dims[-1] // r)
    keys = [k for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad and '.key.' not in k]
    states = [dict() for _ in keys]
    fwd = O.segtran2d_forward if c['dim'] == 2 else O.segtran3d_forward

    def one():
        t0 = time.time()
        for k in keys:
            sd[k].grad = None
        y = fwd(sd, x, dims, training=True)
        loss = O.seg_loss(y, mask, pw)[0]
        loss.backward()
        grads = [sd[k].grad for k in keys]
        with torch.no_grad():
            O.global_clip_([g for g in grads if g is not None], 0.1)
            O.bertadam_step([sd[k] for k in keys], grads, states, 2e-4, [1e-5 if 'backbone' in k else 1e-4 for k in keys], 0.05, 10000)
        return time.time() - t0
    t_start = time.time()
    warm = one()
    times = []
    while len(times) < n_timed and (not times or time.time() - t_start + times[-1] < budget_s):
        times.append(one())
    med = statistics.median(times)
    return {'value': round(batch / med, 4), 'unit': 'images/s' if c['dim'] == 2 else 'volumes/s', 'batch': batch, 'steps_timed': len(times),
            's_per_step_median': round(med, 3), 's_per_step_all': [round(t, 3) for t in times], 's_warmup_step': round(warm, 3)}


def cpu_baseline(cfg_name, threads):
    """SURVEY.md 8(d) / BASELINE.md section 4: the metric configuration at batch 1 (the `value`), plus cfg1 at its full batch (>= 3 warm steps,
    median) and the other family's batch-1 step (cfg4 when the metric is 2-D).  ~60-90 s of CPU work in a child process with a hard limit."""
    main = _cpu_steps(cfg_name, 1, threads, 3, 45.0)
    out = {'value': main['value'], 'unit': main['unit'], 'cores': threads, 'kind': 'port',
           'sample': 'median of %d warm full train steps (fwd+loss+bwd+clip+BertAdam, one untimed warm-up step before) of oracle/segtran_oracle.py '
                     'at batch 1 of the %s shapes and synthetic weights, dropout-free: %.2f s/step (warm-up step %.2f s)'
                     % (main['steps_timed'], cfg_name, main['s_per_step_median'], main['s_warmup_step']),
           'detail': {cfg_name + '_bs1': main}}
    try:
        from segtran_amd import engine
        out['detail']['cfg1_bs%d' % engine.CONFIGS['cfg1']['bs']] = _cpu_steps('cfg1', engine.CONFIGS['cfg1']['bs'], threads, 3, 30.0)
        other = 'cfg4' if engine.CONFIGS[cfg_name]['dim'] == 2 else 'cfg2'
        out['detail'][other + '_bs1'] = _cpu_steps(other, 1, threads, 2, 40.0)
    except Exception as e:                                       # the secondary samples must never cost the primary one
        out['detail']['error'] = repr(e)[:160]
    out['reference_code_8core_cfg1'] = '0.43 images/s (the reference itself on the survey container, BASELINE.md section 3)'
    return out


def run_cpu_baseline(cfg_name, limit_s=420):
    """The CPU leg runs in a child process with a hard time limit so that it can never cost the bench line.
    Threads: physical cores, capped at 64 (PyTorch CPU ops stop scaling -- and oversubscribe -- beyond that)."""
    import subprocess
    threads = max(1, min(64, (os.cpu_count() or 2) // 2))
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', cfg_name, '--threads', str(threads)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=limit_s,
                             env=dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads)))
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        return {'value': None, 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'sample': 'failed: %s' % repr(e)[:160]}


def engine_roofline(prof, steps, cfg_name, engine_name, mode_batch=None):
    """Roofline of the dominant kernel family: every launch of the tile engine inside the timed region is bracketed by HIP events on the
    launch stream (segx.SegxLib.gemm / _timed); achieved = algorithmic FLOPs of those launches / their summed durations."""
    traffic = tnote = None
    tfile = os.path.join(ROOT, 'profiles', 'pmc_engine_traffic.json')
    if os.path.exists(tfile):      # PMC passes cannot run inside the timed bench: read the committed result of the same workload
        tj = json.load(open(tfile)).get(cfg_name if engine_name == 'x6' else cfg_name + '_f32')
        if tj:
            traffic = round(tj['traffic_bytes_per_launch'])
            tnote = 'bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, profiles/' + tj['file']
    alg_bytes = 0.0
    for pr in prof:
        shp = pr[3]
        if isinstance(shp[0], str):                             # conv3d: (tag, M, N, K, B, ...): weights + activation + output
            _, M_, N_, K_, B_ = shp[:5]
            alg_bytes += 4.0 * (M_ * K_ + B_ * (N_ * K_ / 27.0 + M_ * N_)) if shp[0] in ('conv3d_fwd', 'conv3d_halo_fwd') else 4.0 * B_ * (M_ * K_ + N_ * K_ / 27.0 + M_ * N_)
        else:
            M_, N_, K_, nb_ = shp[:4]
            alg_bytes += 4.0 * nb_ * (M_ * K_ + N_ * K_ + M_ * N_)
    sel = {'x6': [p for p in prof if p[4]], 'f32': [p for p in prof if not p[4]]}
    stat = {}
    for k, ps in sel.items():
        fl, ms = sum(p[2] for p in ps), sum(p[0].elapsed_time(p[1]) for p in ps)
        stat[k] = dict(launches_per_step=len(ps) // max(1, steps), ms_per_step=round(ms / max(1, steps), 2),
                       tflop_per_step=round(fl / max(1, steps) / 1e12, 3), tflops=round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0)
    flops = sum(p[2] for p in prof)
    ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    if engine_name == 'x6':
        dom = stat['x6']
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
        roof = {'bound': 'mfma',
                'kernel': 'segx bf16x6 tile engine: gemm_x6* + conv3d_{fwd,wgrad}_x6 + conv3d_halo_{fwd,wgrad}_x6 kernels (fp32 operands split in registers into 3 bf16 '
                          'planes, 6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block: fp32-equivalent results)',
                'achieved': dom['tflops'], 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(dom['tflops'] / peak, 4),
                'peak_note': 'fp32-equivalent TFLOP/s: bf16 dense MFMA peak 2500 / 6 instructions per block product; executed bf16 MFMA rate = '
                             '6 x achieved = %.0f TFLOP/s of 2500' % (6 * dom['tflops']),
                'frac_of_f32_mfma_peak': round(dom['tflops'] / PEAK_F32_MFMA_TFLOPS, 4)}
    else:
        dom = stat['f32']
        roof = {'bound': 'mfma', 'kernel': 'segx fp32-MFMA tile engine: gemm_f32_kernel + conv3d_{fwd,wgrad}_kernel (v_mfma_f32_32x32x2_f32)',
                'achieved': dom['tflops'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(dom['tflops'] / PEAK_F32_MFMA_TFLOPS, 4)}
    # per-shape figures of the dominant engine (the driver's line witnesses the per-kernel claims): the twelve shapes with the
    # most time, plus the squeezed-attention score / value GEMMs (mode-batched: batch = 4 modes x samples, one side = the attractors)
    agg = {}
    for pr in (sel['x6'] if engine_name == 'x6' else sel['f32']):
        a = agg.setdefault(pr[3], [0, 0.0, 0.0]); a[0] += 1; a[1] += pr[0].elapsed_time(pr[1]); a[2] += pr[2]
    peak_ = PEAK_BF16_MFMA_TFLOPS / 6.0 if engine_name == 'x6' else PEAK_F32_MFMA_TFLOPS

    def shape_bytes(shp):
        """algorithmic bytes of ONE launch of this shape (operands read once, result written once; an operand shared by the batch counted once)"""
        if isinstance(shp[0], str):
            _, M_, N_, K_, B_ = shp[:5]
            return 4.0 * (M_ * K_ + B_ * (N_ * K_ / 27.0 + M_ * N_)) if shp[0] in ('conv3d_fwd', 'conv3d_halo_fwd') else 4.0 * B_ * (M_ * K_ + N_ * K_ / 27.0 + M_ * N_)
        M_, N_, K_, nb_ = shp[:4]
        small, big = min(M_ * K_, N_ * K_), max(M_ * K_, N_ * K_)
        return 4.0 * (nb_ * (big + M_ * N_) + (small if nb_ > 1 and small * 8 <= big else nb_ * small))      # a much smaller operand = the shared weights

    def row(shp, v):
        n, t, fl = v
        tf = fl / (t * 1e-3) / 1e12 if t > 0 else 0.0
        by = shape_bytes(shp)
        ai = (fl / n) / by                                                  # FLOP per algorithmic byte
        roof = min(peak_, ai * PEAK_HBM_TBS)                                # TFLOP/s the shape can reach: matrix pipe or HBM, whichever binds first
        r = {'launches_per_step': round(n / max(1, steps), 2), 'ms_per_step': round(t / max(1, steps), 3), 'tflops': round(tf, 1), 'frac': round(tf / peak_, 3),
             'flop_per_byte': round(ai, 1), 'bound': 'mfma' if roof >= peak_ else 'hbm', 'roof_tflops': round(roof, 1), 'frac_of_roof': round(tf / roof, 3),
             'gbs': round(by * n / (t * 1e-3) / 1e9) if t > 0 else 0}
        if isinstance(shp[0], str):
            r.update(kind=shp[0], M=shp[1], N=shp[2], K=shp[3], batch=shp[4])
        else:
            r.update(kind='gemm', M=shp[0], N=shp[1], K=shp[2], batch=shp[3], a_kcontig=bool(shp[4]), b_kcontig=bool(shp[5]), splitk=shp[6], tile=shp[7])
        return r
    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    by_shape = [row(k, v) for k, v in ranked[:12]]
    attn = [row(k, v) for k, v in ranked if not isinstance(k[0], str) and mode_batch and k[3] == mode_batch][:10]
    # how much of the dominant engine's time sits in launches whose ALGORITHMIC intensity puts them under the HBM roof, not the matrix pipe's (the
    # 24..272-channel pointwise convolutions of the backbone at 512 x 512 / 256 x 256 planes): `frac` above prices them against the matrix pipe all the same
    t_all = sum(v[1] for v in agg.values())
    t_hbm = sum(v[1] for k, v in agg.items() if min(peak_, (v[2] / v[0]) / shape_bytes(k) * PEAK_HBM_TBS) < peak_)
    roof['hbm_bound_share_of_engine_time'] = round(t_hbm / t_all, 3) if t_all > 0 else 0.0
    roof['time_weighted_frac_of_applicable_roof'] = round(sum(v[1] * ((v[2] / (v[1] * 1e-3) / 1e12) / min(peak_, (v[2] / v[0]) / shape_bytes(k) * PEAK_HBM_TBS))
                                                              for k, v in agg.items() if v[1] > 0) / t_all, 3) if t_all > 0 else 0.0
    roof.update({'by_shape': by_shape, 'attention_gemms': attn, 'traffic': traffic, 'traffic_note': tnote, 'algorithmic_bytes_per_launch': round(alg_bytes / max(1, len(prof))),
                 'launches_per_step': dom['launches_per_step'], 'gemm_ms_per_step': dom['ms_per_step'], 'gemm_tflop_per_step': dom['tflop_per_step'],
                 'all_engine_launches': {'launches_per_step': len(prof) // max(1, steps), 'ms_per_step': round(ms / max(1, steps), 2),
                                         'tflop_per_step': round(flops / max(1, steps) / 1e12, 3), 'tflops': round(achieved, 2)},
                 'f32_engine_remainder' if engine_name == 'x6' else 'x6_engine_part': stat['f32' if engine_name == 'x6' else 'x6']})
    return roof, achieved


def measure(cfg_name, args, steps, warmup, rank, world, dev, other_order=True, batch=None):
    """W warm-up steps, EXACTLY `steps` timed steps inside a barrier + synchronize bracket (max over ranks), engine launches profiled."""
    from segtran_amd import engine, segx, dist as sdist, functional as SF
    from segtran_amd.networks import segtran_shared as ss
    from segtran_amd.efficientnet.model import MBConvBlock
    from segtran_amd.networks.aj_i3d.aj_i3d import InceptionModule
    c = engine.CONFIGS[cfg_name]
    B = batch or c['bs']
    torch.manual_seed(1234)
    SF.manual_seed(1234 + rank)

    def set_op_order(net_, reassociated):
        # exact re-associations of consecutive linear maps (same function, same parameter gradients): attention projections applied
        # after the contraction with the attractor-side operand, class projection composed into the out-FPN bridge weights
        ss.CrossAttFeatTrans.reassociate_projections = reassociated
        MBConvBlock.gate_in_weights = reassociated                 # squeeze-excite gate folded into the projection weights
        InceptionModule.fuse_reductions = reassociated             # 3-D: the two 1x1x1 reductions of an Inception module as one convolution + one BatchNorm
        net_.fuse_output_tail = reassociated
        if hasattr(net_, 'fuse_input_bridge'):
            net_.fuse_input_bridge = reassociated          # 3-D: in_bridge_to3 composed into the stem filters

    net = engine.build_model(cfg_name, dev)
    set_op_order(net, not args.reference_op_order)
    sdist.enable_sync_batchnorm()
    net.train()
    opt = engine.init_optimizer(net, c['task'])
    reducer = sdist.GradReducer(opt) if world > 1 else None
    step = engine.TrainStep(net, opt, c['task'], reducer)
    x, raw = engine.synth_batch(cfg_name, B, dev, seed=1337 + rank)          # disjoint samples per rank
    L = segx.lib()
    graphed = bool(args.graph) and world == 1
    if graphed:                         # the whole step as ONE hipGraph launch (engine.GraphedTrainStep); per-launch engine events are not available
        step = engine.GraphedTrainStep(step, x, raw, warmup=min(3, max(1, warmup)))
    for _ in range(warmup):
        step(x, raw)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # HIP events bracket every engine launch of EVERY prof_every-th timed step (two event records per launch cost host time and a queue marker each:
    # r04_w measured the same step at 67.2 ms without them -- H2D copies included -- and 68.9 ms with them on every step); the roofline is computed
    # over those launches (5 of 50 steps by default; every step when fewer than 10 steps are timed: profiler runs)
    prof_every = 1 if (steps < 10 or graphed) else PROF_EVERY
    prof_list = None if graphed else []
    prof_steps = 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    # the per-launch profiling events are ~700 Python objects per step: a generation-2 collection in the middle of the timed region stalls the host for
    # tens of milliseconds (r03_af / r03_ak: one 116..120-ms step among fifty 75-ms ones).  Collect now, keep the collector out of the timed region.
    import gc
    gc.collect(); gc.disable()
    try:
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            on = prof_list is not None and i % prof_every == 0
            L.gemm_prof = prof_list if on else None
            prof_steps += 1 if on else 0
            loss = step(x, raw)
            ev[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    prof, L.gemm_prof = (prof_list or []), None
    per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    other = None
    with_h2d = None
    if world == 1 and not graphed and not args.single_order:
        # SURVEY.md 8(d) starts the step at the H2D copy of a pre-generated batch: the same step fed from a PINNED host batch, copy on the step's
        # stream inside the timed region (no loader thread hiding it).  Reported beside `value`, which keeps the inputs resident in HBM.
        xh, rh = x.cpu().pin_memory(), raw.cpu().pin_memory()
        kk = max(3, min(10, steps // 4))
        for i in range(kk + 2):
            if i == 2:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            x.copy_(xh, non_blocking=True); raw.copy_(rh, non_blocking=True)
            step(x, raw)
        torch.cuda.synchronize()
        hd = (time.perf_counter() - t1) / kk
        with_h2d = {'value': round(B / hd, 3), 'ms_per_step': round(hd * 1e3, 2), 'steps': kk, 'batch_bytes': xh.numel() * xh.element_size() + rh.numel() * rh.element_size(),
                    'note': 'pinned host batch copied H2D on the step stream inside the timed region (PCIe-inclusive rate; not `value`)'}
    if graphed:
        step.close()
    if world == 1 and other_order and not args.single_order and not graphed:   # the same step in the OTHER operation order, outside the timed region
        set_op_order(net, args.reference_op_order)
        for _ in range(2):
            step(x, raw)
        torch.cuda.synchronize()
        k = max(2, min(10, steps // 4))
        t1 = time.perf_counter()
        for _ in range(k):
            step(x, raw)
        torch.cuda.synchronize()
        other = (time.perf_counter() - t1) / k
        set_op_order(net, not args.reference_op_order)
    lossv = float(loss.detach())
    assert lossv == lossv, 'loss is NaN'
    L.team_check()                                                 # a timed-out team exchange anywhere in the run voids the number
    overlap = None
    if reducer is not None:
        overlap = reducer.overlap_stats()                          # buckets whose all-reduce was launched from inside backward (of the last step)
        reducer.close()
    del step, opt, net, reducer
    torch.cuda.empty_cache()
    unit = 'images/s' if c['dim'] == 2 else 'volumes/s'
    roof, achieved = engine_roofline(prof, max(1, prof_steps), cfg_name, args.engine, mode_batch=4 * B) if (rank == 0 and prof) else (None, 0.0)
    if roof:
        roof['profiled_steps'] = '%d of %d timed steps (every %d%s)' % (prof_steps, steps, prof_every, 'th' if prof_every > 1 else '')
    if rank == 0 and os.environ.get('SEGX_BENCH_VERBOSE'):
        agg = {}
        for e0, e1, fl, shp, x6 in prof:
            a = agg.setdefault(shp + (('x6',) if x6 else ('f32',)), [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
        print('[bench] %s engine launches by time (M,N,K,batch,A_kcontig,B_kcontig,splitk,tile,engine): count ms TFLOP/s' % cfg_name, file=sys.stderr)
        top = None if os.environ['SEGX_BENCH_VERBOSE'] == '2' else 40
        for shp, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            print('[bench]   %-58s %4d %8.2f %7.1f' % (shp, n, t, fl / (t * 1e-3) / 1e12), file=sys.stderr)
    med = statistics.median(per_step)
    res = {'metric': 'train-step %s (%s, %s op order)' % (unit.replace('/s', '/sec'), cfg_name, 'reference' if args.reference_op_order else 're-associated'), 'value': round(world * B * steps / dt, 3), 'unit': unit,
           'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 2),
           'ms_per_step_median': round(med, 2), 'value_at_median': round(world * B / (med * 1e-3), 3),
           'ms_per_step_min_max': [round(min(per_step), 2), round(max(per_step), 2)],
           'config': {'workload': WORKLOADS.get(cfg_name, cfg_name), 'global_batch': world * B, 'per_gpu_batch': B, 'parallelism': 'dp%d' % world,
                      'dropout': 0.2, 'step': 'fwd+BCE/Dice+bwd+allreduce+clip+BertAdam' + (' (one hipGraph launch per step)' if graphed else ''), 'final_loss': round(lossv, 5),
                      'gemm_path': 'bf16x6 tile engine (fp32-equivalent; float4-legal operands with > 48 rows per side), fp32 MFMA for the rest'
                                   if args.engine == 'x6' else 'fp32 MFMA tile engine',
                      'op_order': ('reference' if args.reference_op_order else 're-associated') + ' (DESIGN.md 5b: exact re-association of '
                                  'consecutive linear maps; every layer, parameter and gradient is computed)',
                      ('reassociated_op_order' if args.reference_op_order else 'reference_op_order'):
                          None if other is None else {'value': round(B / other, 3), 'ms_per_step': round(other * 1e3, 2)},
                      'with_h2d_copy': with_h2d, 'overlap': overlap,
                      'ranks': world, 'collective_backend': ((torch.distributed.get_backend() + (' (RCCL)' if torch.distributed.get_backend() == 'nccl' else '')) if world > 1 else None)},
           'roofline': roof}
    if rank == 0:
        print('[bench] %s: %.1f ms/step (median %.1f), %.2f %s, engine %.1f TFLOP/s' % (cfg_name, res['ms_per_step'], med, res['value'], unit, achieved),
              file=sys.stderr, flush=True)
    return res


def run_graph_replay(args, config=None, graph=True, batch=0):
    """The same configuration replayed as ONE captured hipGraph per step (engine.GraphedTrainStep), in a child process after everything else has been
    measured: a capture problem can then cost this extra block only, never the line.  Not the headline: `value` stays the eager step, whose engine
    launches carry the HIP events the roofline is computed from."""
    # an EAGER launch-bound step (cfg1, one image per rank) is timed by the host, and the host needs ~30 steps of a fresh process to reach its pace (r06_bn: 22.5 -> 17.2 -> 15.3 ms over the
    # first three blocks of ten cfg1 steps, then flat): HOST_WARMUP untimed steps there; graph replays and device-bound steps keep the short warm-up
    warm = max(3, args.warmup // 2) if graph else max(HOST_WARMUP, args.warmup)
    cmd = [sys.executable, os.path.abspath(__file__)] + (['--graph'] if graph else []) + ['--config', config or args.config, '--engine', args.engine, '--steps', str(max(10, args.steps // 2)),
           '--warmup', str(warm), '--no-brats', '--no-cpu-baseline', '--single-order'] + (['--batch', str(batch)] if batch else [])
    if args.reference_op_order:
        cmd.append('--reference-op-order')
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd=ROOT)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
        if out.returncode != 0 or not line:
            return {'error': 'rc %d: %s' % (out.returncode, out.stderr.decode()[-200:])}
        r = json.loads(line[-1])
        return {k: r.get(k) for k in ('value', 'unit', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median')}
    except subprocess.TimeoutExpired:
        return {'error': 'timeout'}


HOST_WARMUP = 40            # untimed steps before a host-bound (eager, launch-bound) configuration is timed: see run_graph_replay
LINE_LIMIT = 4096           # the driver keeps an ~8 KB tail of stdout (BENCH_r04: a 24.9 KB line arrived without its head): the line stays under half of that
_ATTN_NAMES = ('QK^T', 'P.V', 'dP', 'dV', 'P.(vW)', 'dS.K')


def _short_roofline(roof, n_shapes=6):
    """The scalars of a roofline block plus at most n_shapes attention GEMMs as compact arrays [M, N, K, batch, ms_per_step, TFLOP/s, frac]."""
    if not roof:
        return None
    out = {k: roof.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
    out['kernel'] = 'segx bf16x6 tile engine (gemm_x6* + conv3d_*_x6 + conv3d_halo_*_x6 kernels)' if 'bf16x6' in roof.get('kernel', '') else 'segx fp32-MFMA tile engine'
    out['algorithmic_bytes'] = roof.get('algorithmic_bytes_per_launch')
    out['launches'] = roof.get('launches_per_step')
    out['ms_per_step'] = roof.get('gemm_ms_per_step')
    out['traffic_src'] = (roof.get('traffic_note') or '').split('profiles/')[-1] or None
    out['profiled_steps'] = roof.get('profiled_steps')
    out['f32_remainder_ms'] = (roof.get('f32_engine_remainder') or roof.get('x6_engine_part') or {}).get('ms_per_step')
    rows = roof.get('attention_gemms') or []
    out['attention_gemms'] = {'cols': ['M', 'N', 'K', 'batch', 'ms_per_step', 'tflops', 'frac'],
                              'rows': [[r['M'], r['N'], r['K'], r['batch'], r['ms_per_step'], r['tflops'], r['frac']] for r in rows[:n_shapes]]}
    return out


def compact_line(res):
    """The ONE JSON line rank 0 prints: headline fields, `config`, the scalars of `roofline`, `cpu_baseline` (value + medians), the BraTS / polyp blocks as
    value / ms_per_step / roofline.frac, and <= 6 attention GEMM rows.  Everything else (the per-shape tables, notes, per-step lists) goes to
    bench_shapes.json beside this file.  tests/test_bench_line.py holds the size bound."""
    c = res.get('config') or {}

    def ms(d):
        return None if not isinstance(d, dict) else (d.get('ms_per_step') if 'error' not in d else 'error')
    other_key = 'reassociated_op_order' if 'reassociated_op_order' in c else 'reference_op_order'
    line = {k: res.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median', 'higher_is_better', 'scaling',
                                    'vs_baseline', 'dtype', 'data')}
    line['config'] = {'workload': c.get('workload'), 'global_batch': c.get('global_batch'), 'per_gpu_batch': c.get('per_gpu_batch'),
                      'parallelism': c.get('parallelism'), 'dropout': c.get('dropout'), 'step': c.get('step'), 'final_loss': c.get('final_loss'),
                      'op_order': (c.get('op_order') or '').split(' (')[0], other_key + '_ms': ms(c.get(other_key)),
                      other_key + '_value': (c.get(other_key) or {}).get('value') if isinstance(c.get(other_key), dict) else None,      # the like-for-like figure beside `value`
                      'with_h2d_copy_ms': ms(c.get('with_h2d_copy')), 'with_h2d_copy_value': (c.get('with_h2d_copy') or {}).get('value') if isinstance(c.get('with_h2d_copy'), dict) else None, 'hipgraph_replay_ms': ms(c.get('hipgraph_replay')),
                      'cfg1_hipgraph_ms': ms(c.get('hipgraph_replay_cfg1')), 'cfg1_eager_ms': ms(c.get('eager_cfg1')),
                      'ranks': c.get('ranks'), 'collective_backend': c.get('collective_backend'), 'overlap': c.get('overlap')}
    line['roofline'] = _short_roofline(res.get('roofline'))
    cb = res.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = {'value': cb.get('value'), 'unit': cb.get('unit'), 'cores': cb.get('cores'), 'kind': cb.get('kind'),
                                'sample': (cb.get('sample') or '')[:200],
                                's_per_step_median': {k: v.get('s_per_step_median') for k, v in (cb.get('detail') or {}).items() if isinstance(v, dict)}}
    for blk in ('brats', 'polyp'):
        if res.get(blk):
            line[blk] = {}
            for k, r in res[blk].items():
                b = {'value': r.get('value'), 'unit': r.get('unit'), 'ms_per_step': r.get('ms_per_step'), 'ms_per_step_median': r.get('ms_per_step_median'),
                     'steps': r.get('steps'), 'per_gpu_batch': (r.get('config') or {}).get('per_gpu_batch')}
                if r.get('roofline'):
                    rr = _short_roofline(r['roofline'], n_shapes=3)
                    b['roofline'] = {kk: rr[kk] for kk in ('achieved', 'peak', 'frac', 'traffic', 'algorithmic_bytes', 'launches', 'ms_per_step')}
                    b['roofline']['attention_gemms'] = rr['attention_gemms']['rows']
                if (r.get('config') or {}).get('overlap'):
                    b['overlap'] = r['config']['overlap']
                line[blk][k] = b
    line['full'] = 'bench_shapes.json'
    s = json.dumps(line, separators=(',', ':'))
    if len(s) > LINE_LIMIT:                 # never let a long note cost the record: drop the optional parts in order until the line fits
        for path in (('roofline', 'attention_gemms'), ('cpu_baseline', 'sample'), ('brats',), ('polyp',)):
            d = line
            for k in path[:-1]:
                d = d.get(k) or {}
            d.pop(path[-1], None)
            s = json.dumps(line, separators=(',', ':'))
            if len(s) <= LINE_LIMIT:
                break
    return s


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher in front (WORLD_SIZE unset): re-run this command line as N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 (what the driver does for N > 1).  Rank 0 of the child job prints the one JSON line; its `n_gpus` and
    `config.ranks` say how many ranks actually ran.  Returns the job's exit code."""
    import socket
    if torch.cuda.is_available() and torch.cuda.device_count() < n and not os.environ.get('SEGX_BENCH_SHARE_GPU'):
        print('bench.py --gpus %d: only %d GPU(s) visible; refusing to report a %d-GPU number from fewer devices' % (n, torch.cuda.device_count(), n),
              file=sys.stderr)
        return 2
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cpu-baseline-only', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--threads', type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='cfg2', help='BASELINE config of the MAIN measurement (cfg2 = metric default)')
    ap.add_argument('--engine', default=os.environ.get('SEGX_ENGINE', 'x6'), choices=['x6', 'f32'],
                    help="tile engine: 'x6' = bf16x6 (default), 'f32' = v_mfma_f32_32x32x2_f32 everywhere")
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch of the MAIN measurement (0 = the configuration\'s own)')
    ap.add_argument('--graph', action='store_true', help='replay the step as one captured hipGraph (single GPU; no per-launch roofline)')
    ap.add_argument('--no-brats', action='store_true', help='skip the secondary BraTS block(s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--single-order', action='store_true', help='skip the comparison run in the other operation order (profiling runs)')
    ap.add_argument('--reference-op-order', action='store_true',
                    help="time the reference's operation order (no linear-chain re-association, DESIGN.md section 5b) as the main number")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.cpu_baseline_only, args.threads)))
        return

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_spawn(args.gpus))               # `python bench.py --gpus N` on its own: start the N ranks here
    from segtran_amd import segx, dist as sdist
    rank, local, world = sdist.init_distributed()
    assert world == max(1, args.gpus), 'WORLD_SIZE %d != --gpus %d' % (world, args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    if world > 1 and torch.cuda.device_count() < world and not os.environ.get('SEGX_BENCH_SHARE_GPU'):
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU; SEGX_BENCH_SHARE_GPU=1 lets test ranks share a device)'
                         % (world, torch.cuda.device_count()))
    dev = torch.device('cuda', local % torch.cuda.device_count() if os.environ.get('SEGX_BENCH_SHARE_GPU') else local)
    torch.cuda.set_device(dev)
    L = segx.lib()
    L.set_engine(args.engine)
    for kv in filter(None, os.environ.get('SEGX_TUNE', '').split(',')):       # e.g. SEGX_TUNE=2:1 (bisecting knobs, see segx_tune)
        k, v = kv.split(':'); L.c.segx_tune(int(k), int(v))

    res = measure(args.config, args, args.steps, args.warmup, rank, world, dev, batch=args.batch or None)
    res = dict(res, higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic')
    if not args.no_brats and args.config == 'cfg2':
        k2, w2 = max(10, args.steps // 2), max(3, args.warmup // 2)
        brats = {}
        for cfg in (['cfg4'] if world == 1 else []) + ['cfg5']:
            r = measure(cfg, args, k2, w2, rank, world, dev, other_order=False)
            brats[cfg] = {k: r[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median',
                                             'value_at_median', 'config', 'roofline')}
        res['brats'] = brats
    if not args.no_brats and args.config == 'cfg2' and world > 1:
        # BASELINE configs[2]: polyp 352 x 352 data-parallel (quoted on 4 GPUs) -- at 6 images per GPU like the other blocks (weak scaling) AND with the
        # reference's semantics, where --bs 6 is the GLOBAL batch split over the ranks (train2d.py:791: 6 // 4 = 1 image per rank, a launch-bound step)
        k3, w3 = max(10, args.steps // 2), max(3, args.warmup // 2)
        polyp = {}
        for tag, bsz in (('cfg3_bs6_per_gpu', 6), ('cfg3_global_bs6', max(1, 6 // world))):
            r = measure('cfg3', args, k3, w3, rank, world, dev, other_order=False, batch=bsz)
            polyp[tag] = {k: r[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median', 'value_at_median', 'config')}
        res['polyp'] = polyp
    if not args.no_brats and args.config == 'cfg2' and world == 1 and not args.graph:
        # r06: BASELINE configs[2] on ONE GPU as well -- 6 images per step, and the reference's per-rank step of `--bs 6` on 4 ranks
        # (train2d.py:791: 6 // 4 = 1 image per rank; launch-bound: eager and replayed as one hipGraph)
        k3, w3 = max(10, args.steps // 2), max(3, args.warmup // 2)
        polyp = {}
        for tag, bsz in (('cfg3_bs6', 6), ('cfg3_bs1_per_rank_of_4', 1)):
            r = measure('cfg3', args, k3, w3 if bsz > 1 else max(HOST_WARMUP, w3), rank, world, dev, other_order=False, batch=bsz)      # one image: host-bound, see HOST_WARMUP
            polyp[tag] = {k: r[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median', 'value_at_median', 'config')}
        res['polyp'] = polyp
    if rank != 0:
        return
    if world == 1 and not args.no_brats and args.config == 'cfg2' and not args.graph and not args.single_order:
        res['polyp']['cfg3_bs1_hipgraph'] = run_graph_replay(args, config='cfg3', batch=1)
    if world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = run_cpu_baseline(args.config)
    if world == 1 and not args.graph and not args.single_order:
        res['config']['hipgraph_replay'] = run_graph_replay(args)
        if args.config == 'cfg2':                      # the launch-bound configuration (256 x 256, bs 2) as one graph: where capture matters most
            res['config']['hipgraph_replay_cfg1'] = run_graph_replay(args, config='cfg1')
            res['config']['eager_cfg1'] = run_graph_replay(args, config='cfg1', graph=False)    # the same step launched eagerly: host time of a small per-rank batch
    try:                                               # the whole result (per-shape tables, notes, per-step lists) beside the line
        json.dump(res, open(os.path.join(ROOT, 'bench_shapes.json'), 'w'), indent=1)
    except OSError:
        pass
    print(compact_line(res), flush=True)


if __name__ == '__main__':
    main()
