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
         '-Wno-unused-value']


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
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (s, out))
        if verbose and out.strip():
            print(out)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    open(stamp, 'w').write(h.hexdigest())
    return lib


if __name__ == '__main__':
    print(build(verbose=True))
