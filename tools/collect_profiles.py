"""Copy the summaries of a tools/gpu_round.sh session into profiles/ under the round tag and derive the MFMA-engine HBM traffic.
python tools/collect_profiles.py <tag>      (build container, after gpurun merged gpurun_out/<tag>/)"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, 'gpurun_out', tag), os.path.join(ROOT, 'profiles')
ENGINE = ('gemm_f32_kernel', 'conv3d_fwd_kernel', 'conv3d_wgrad_kernel')


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
        if 'bn_act_fwd_kernel' in k and k in W:
            cal = (F[k]['FETCH_SIZE']['mean'] * 1024, W[k]['WRITE_SIZE']['mean'] * 1024)
    return {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --config %s --steps 1 --warmup 1`, %s' % (cfg, tag),
            'kernels': ' + '.join(ENGINE), 'launches': n, 'fetch_kb_total_raw': fk, 'write_kb_total': wk,
            'correction': 'gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section): x2'
                          + ('; calibration on bn_act_fwd_kernel (reads == writes): FETCH %.2f MB vs WRITE %.2f MB per launch' % (cal[0] / 1e6, cal[1] / 1e6) if cal else ''),
            'traffic_bytes_per_launch': (2.0 * fk + wk) * 1024.0 / max(1, n)}


traffic = {}
for name in sorted(os.listdir(src)):
    if name.endswith(('.json', '.csv', '.txt', '.log')) and not name.startswith(('prof_', 'pmc_cfg2_FETCH_SIZE.log', 'pmc_cfg2_WRITE_SIZE.log', 'pmc_cfg4')) or name.endswith('_by_kernel.json'):
        shutil.copy(os.path.join(src, name), os.path.join(dst, '%s_%s' % (tag, name)))
for cfg in ('cfg2', 'cfg4'):
    t = engine_traffic(cfg)
    if t:
        json.dump(t, open(os.path.join(dst, '%s_pmc_engine_traffic_%s.json' % (tag, cfg)), 'w'), indent=1)
        traffic[cfg] = {'traffic_bytes_per_launch': t['traffic_bytes_per_launch'], 'file': '%s_pmc_engine_traffic_%s.json' % (tag, cfg)}
if traffic:
    json.dump(traffic, open(os.path.join(dst, 'pmc_engine_traffic.json'), 'w'), indent=1)      # what bench.py reads
print(sorted(n for n in os.listdir(dst) if n.startswith(tag)))
