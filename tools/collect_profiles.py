"""Copy the summaries of a tools/gpu_session.sh (r04) / tools/gpu_round.sh session into profiles/ under the round tag and derive the MFMA-engine HBM traffic.
python tools/collect_profiles.py <tag>      (build container, after gpurun merged gpurun_out/<tag>/)"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, 'gpurun_out', tag), os.path.join(ROOT, 'profiles')
# every kernel whose launches bench.py brackets with events (segx.SegxLib.gemm / conv3d: `roofline.algorithmic_bytes` averages over all of them).  r04 left out
# gemm_x6_lean_kernel ('gemm_x6_kernel' is not a substring of it): 134 of 456 launches at cfg2 were missing from the traffic and matrix-pipe figures.
ENGINE = ('gemm_f32_kernel', 'gemm_skinny_nt_kernel', 'gemm_x6_kernel', 'gemm_x6_lean_kernel', 'gemm_x6ws_kernel', 'gemm_x6ws_pre_kernel', 'gemm_stream_kernel', 'conv3d_fwd_kernel', 'conv3d_wgrad_kernel',
          'conv3d_fwd_x6_kernel', 'conv3d_wgrad_x6_kernel', 'conv3d_halo_fwd_x6_kernel', 'conv3d_halo_wgrad_x6_kernel')


def engine_traffic(cfg):
    """bytes per launch over every launch of the MFMA tile engine: FETCH_SIZE (KB, x2 on gfx950) + WRITE_SIZE (KB)."""
    f = os.path.join(src, 'pmc_%s_FETCH_SIZE_by_kernel.json' % cfg); w = os.path.join(src, 'pmc_%s_WRITE_SIZE_by_kernel.json' % cfg)
    if not (os.path.exists(f) and os.path.exists(w)):
        return None
    F, W = json.load(open(f)), json.load(open(w))
    n = fk = wk = 0
    for k, v in F.items():
        if any(e in k for e in ENGINE):
            n += v['FETCH_SIZE']['launches']; fk += v['FETCH_SIZE']['total']
    for k, v in W.items():
        if any(e in k for e in ENGINE):
            wk += v['WRITE_SIZE']['total']
    cal = None
    for k in F:                                                 # calibration kernel: reads == writes by construction
        if 'gn_apply_kernel' in k and k in W:
            cal = (F[k]['FETCH_SIZE']['mean'] * 1024, W[k]['WRITE_SIZE']['mean'] * 1024)
    return {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --config %s --steps 1 --warmup 1`, %s' % (cfg, tag),
            'kernels': ' + '.join(ENGINE), 'launches': n, 'fetch_kb_total_raw': fk, 'write_kb_total': wk,
            'correction': 'gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section): x2'
                          + ('; calibration on gn_apply_kernel (reads == writes): FETCH %.2f MB vs WRITE %.2f MB per launch' % (cal[0] / 1e6, cal[1] / 1e6) if cal else ''),
            'traffic_bytes_per_launch': (2.0 * fk + wk) * 1024.0 / max(1, n)}


def mfma_busy(cfg):
    """Matrix-pipe utilisation per engine kernel from the SQ counter pass: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip,
    GRBM_GUI_ACTIVE over the 8 XCDs (calibration: SQ_BUSY_CYCLES, summed over 32 shader engines, is 4.0x GRBM_GUI_ACTIVE in every row), so
    busy fraction = MFMA_BUSY / (1024 x GUI_ACTIVE / 8).  Kernels are serialised under counter collection; clocks differ from the timed run."""
    f = os.path.join(src, 'pmc_%s_SQ_VALU_MFMA_BUSY_CYCLES_by_kernel.json' % cfg)
    if not os.path.exists(f):
        return None
    out, tot_m, tot_g = {}, 0.0, 0.0
    for k, v in json.load(open(f)).items():
        if not any(e in k for e in ENGINE) or 'GRBM_GUI_ACTIVE' not in v:
            continue
        m, g = v['SQ_VALU_MFMA_BUSY_CYCLES']['total'], v['GRBM_GUI_ACTIVE']['total']
        wc = max(v['SQ_WAVE_CYCLES']['total'], 1.0)
        out[k] = {'launches': v['GRBM_GUI_ACTIVE']['launches'], 'mfma_busy_frac': round(m / (128.0 * g), 4),
                  'sq_busy_over_gui': round(v['SQ_BUSY_CYCLES']['total'] / g, 3),
                  'wave_cycles_waiting_any': round(v['SQ_WAIT_ANY']['total'] / wc, 3), 'wave_cycles_issue_stalled': round(v['SQ_WAIT_INST_ANY']['total'] / wc, 3),
                  'wave_cycles_issuing': round(v['SQ_ACTIVE_INST_ANY']['total'] / wc, 3)}
        if 'x6' in k:
            tot_m += m; tot_g += g
    return {'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU '
                      'GRBM_GUI_ACTIVE of `bench.py --config %s --steps 1 --warmup 1`, %s' % (cfg, tag),
            'bf16x6_engine_mfma_busy_frac': round(tot_m / (128.0 * tot_g), 4) if tot_g else None, 'per_kernel': out}


traffic = {}
for name in sorted(os.listdir(src)):
    if name.endswith('_kernel_trace.csv'):
        continue                                                  # per-dispatch traces (1 MB each) stay in gpurun_out/
    if name.endswith(('.json', '.csv', '.txt', '.log')) and not name.startswith(('prof_', 'pmc_')) and not name.endswith('.log') or name.endswith('_by_kernel.json'):
        shutil.copy(os.path.join(src, name), os.path.join(dst, '%s_%s' % (tag, name)))
for cfg in ('cfg2', 'cfg4', 'cfg5'):
    t = engine_traffic(cfg)
    if t:
        json.dump(t, open(os.path.join(dst, '%s_pmc_engine_traffic_%s.json' % (tag, cfg)), 'w'), indent=1)
        traffic[cfg] = {'traffic_bytes_per_launch': t['traffic_bytes_per_launch'], 'file': '%s_pmc_engine_traffic_%s.json' % (tag, cfg)}
    b = mfma_busy(cfg)
    if b:
        json.dump(b, open(os.path.join(dst, '%s_mfma_busy_%s.json' % (tag, cfg)), 'w'), indent=1)
if traffic:
    json.dump(traffic, open(os.path.join(dst, 'pmc_engine_traffic.json'), 'w'), indent=1)      # what bench.py reads
print(sorted(n for n in os.listdir(dst) if n.startswith(tag)))
