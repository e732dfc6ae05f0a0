"""Build segtran_amd/lib/libsegx.so from segtran_amd/csrc/*.hip with hipcc for gfx950 (in-tree, no JIT cache).

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot.
"""
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'lib')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-mllvm', '-pragma-unroll-threshold=200000',
         '-Wno-unused-value', '-Rpass-analysis=kernel-resource-usage']       # the remarks: registers / scratch / LDS / occupancy per kernel -> lib/resource_usage.json


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, '*.hip')))
    deps = srcs + sorted(glob.glob(os.path.join(SRC, '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'segx.h')]
    h = hashlib.sha1(' '.join(FLAGS).encode())
    for d in deps:
        h.update(open(d, 'rb').read())
    lib, stamp = os.path.join(OUT, 'libsegx.so'), os.path.join(OUT, 'libsegx.stamp')
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return lib
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        procs.append((s, subprocess.Popen([hipcc] + FLAGS + ['-c', s, '-o', o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    usage = {}
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (s, out))
        usage.update(parse_resource_usage(out, os.path.basename(s)))
        if verbose:
            rest = '\n'.join(l for l in out.splitlines() if 'remark:' not in l)
            if rest.strip():
                print(rest)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    import json
    json.dump(usage, open(os.path.join(OUT, 'resource_usage.json'), 'w'), indent=0, sort_keys=True)
    open(stamp, 'w').write(h.hexdigest())
    return lib


def parse_resource_usage(text, src):
    """hipcc -Rpass-analysis=kernel-resource-usage -> {mangled kernel name: {file, vgprs, agprs, sgprs, scratch, lds, occupancy}} (scratch in bytes per lane:
    non-zero = the register allocator spilled; tests/test_abi.py holds every shipped kernel to zero)"""
    import re
    out = {}
    for blk in re.split(r'remark: [^\n]*Function Name: ', text)[1:]:
        name = blk.split()[0]

        def g(key):
            m = re.search(key + r': (\d+)', blk)
            return int(m.group(1)) if m else -1
        out[name] = {'file': src, 'vgprs': g('VGPRs'), 'agprs': g('AGPRs'), 'sgprs': g('SGPRs'), 'scratch': g(r'ScratchSize \[bytes/lane\]'),
                     'lds': g(r'LDS Size \[bytes/block\]'), 'occupancy': g(r'Occupancy \[waves/SIMD\]')}
    return out


if __name__ == '__main__':
    print(build(verbose=True))
