"""CPU: the C ABI contract -- header, ctypes signatures and exported symbols agree."""
import os, re, subprocess
import pytest
from segtran_amd import segx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, 'include', 'segx.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|int64_t)\s+(segx_\w+)\s*\(([^;]*?)\)\s*;', hdr, flags=re.S):
        name, params = m.group(1), [p.strip() for p in m.group(2).split(',')]
        sig = ''
        for prm in params:
            if prm == 'void':
                continue
            if '*' in prm: sig += 'p'
            elif prm.startswith('int64_t'): sig += 'l'
            elif prm.startswith('uint64_t'): sig += 'u'
            elif prm.startswith('float'): sig += 'f'
            elif prm.startswith('int'): sig += 'i'
            else: sig += '?'
        out[name] = sig
    return out


def test_ctypes_signatures_match_header():
    decl = _declared()
    for name, sig in segx._SIGS.items():
        assert name in decl, name + ' missing from include/segx.h'
        assert decl[name] == sig, '%s: header %s vs binding %s' % (name, decl[name], sig)
    extra = set(decl) - set(segx._SIGS) - {'segx_version', 'segx_last_error', 'segx_gemm_f32', 'segx_gemm_plan', 'segx_gemm_plan_model'}
    assert not extra, 'declared but unbound: %s' % sorted(extra)


def test_hip_library_exports_every_declared_symbol():
    """The in-tree HIP build must load and export the whole ABI (no compute call: there is no GPU here)."""
    from segtran_amd.build import build
    lib = build()
    syms = subprocess.check_output(['nm', '-D', '--defined-only', lib]).decode()
    exported = set(re.findall(r' T (segx_\w+)', syms))
    assert set(_declared()) <= exported, sorted(set(_declared()) - exported)
    L = segx.SegxLib(lib)
    assert L.c.segx_version() >= 100 and not L.emulated


def test_product_refuses_cpu_tensors_and_missing_library(tmp_path):
    import torch
    from segtran_amd.build import build
    L = segx.SegxLib(build())
    with pytest.raises(RuntimeError, match='CPU tensor'):
        L.layernorm_fwd(torch.zeros(4, 8), None, None, torch.zeros(4, 8), torch.zeros(4), torch.zeros(4), 4, 8, 1e-12)
    with pytest.raises(RuntimeError, match='not built'):
        segx.SegxLib(str(tmp_path / 'nope.so'))


def test_gemm_desc_layout_matches_header():
    """ctypes mirror of segx_gemm_desc: same fields in the same order as include/segx.h (a silent mismatch would shift every later field)."""
    hdr = open(os.path.join(ROOT, 'include', 'segx.h')).read()
    body = re.search(r'typedef struct \{(.*?)\} segx_gemm_desc;', hdr, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for stmt in body.split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = re.sub(r'^(const\s+)?(int32_t|int64_t|uint64_t|float|void)\s*\*?', '', stmt)
        names += [n.strip().lstrip('*').strip() for n in decl.split(',')]
    assert names == [f[0] for f in segx.GemmDesc._fields_], (names, [f[0] for f in segx.GemmDesc._fields_])


def test_product_library_rejects_the_ablation_variants():
    """include/segx.h: knob 6 values 2..5 (kernels whose results are NOT the GEMM) exist in -DSEGX_BENCH builds only."""
    from segtran_amd.build import build
    L = segx.SegxLib(build())
    for v in (2, 3, 4, 5):
        assert L.c.segx_tune(6, v) == -1
    assert L.c.segx_tune(6, 1) == 0 and L.c.segx_tune(6, 0) == 0
    # knobs reject what the product suite does not cover: unknown knob ids and out-of-range settings
    for knob, v in ((1, 3), (2, 2), (7, 3), (8, 10), (9, 12), (10, 16), (10, 8), (11, 0)):
        assert L.c.segx_tune(knob, v) == -1, (knob, v)


def test_no_shipped_kernel_spills_registers_beyond_the_known_gemm_tails():
    """`Scratch_Size` of the shipped kernels.  segtran_amd/build.py compiles with -Rpass-analysis=kernel-resource-usage and keeps the
    per-kernel figures in lib/resource_usage.json (same build as the .so: the stamp covers both).  Every kernel outside gemm.hip must have NO scratch
    (r04: team / resident BatchNorm forms spilled 52-312 bytes per lane); the tile-engine kernels listed below spill a few registers around their
    epilogues (outside the k-loop) -- the list is closed: a new spill, or a growing one, fails here."""
    import json
    from segtran_amd.build import build, OUT
    build()
    usage = json.load(open(os.path.join(OUT, 'resource_usage.json')))
    assert len(usage) > 400                                   # every __global__ instantiation of the library reports
    known = {'gemm_x6ws_kernel': 52, 'gemm_f32_kernel': 36, 'gemm_x6_kernel': 20, 'gemm_x6_lean_kernel': 20}
    # r06: the 6-wave (192 output channels) form of the resident-halo weight gradient sits at the 168 registers of three waves per SIMD (112 of them accumulators) and keeps
    # ten loop-invariant staging offsets in scratch (reloaded once per 42-MFMA tile); the 4-wave form and the forward kernels have none
    known_other = {('conv3d_halo.hip', 'conv3d_halo_wgrad_x6_kernelILi6E'): 48}
    bad = []
    for name, u in usage.items():
        assert u['scratch'] >= 0 and u['vgprs'] > 0, name
        if u['scratch'] == 0:
            continue
        fam = [k for k in known if ('4segx%d%sI' % (len(k), k)) in name]
        if any(u['file'] == f and k in name and u['scratch'] <= cap for (f, k), cap in known_other.items()):
            continue
        if u['file'] != 'gemm.hip' or not fam or u['scratch'] > known[fam[0]]:
            bad.append((name, u['file'], u['scratch']))
    assert not bad, bad
    # the team BatchNorm kernels keep >= 2 workgroups per CU resident (the forward-progress argument of common.h: team_exchange counts on it)
    team = [u for n, u in usage.items() if 'bn_act_fwd_team_kernel' in n or 'bn_act_bwd_team_kernel' in n]
    assert team and all(u['occupancy'] >= 2 and u['scratch'] == 0 for u in team)
