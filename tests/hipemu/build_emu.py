"""Build tests/hipemu/build/libsegx_emu.so: the product's HIP sources compiled UNMODIFIED by the host
clang++ against the fiber SIMT emulator in tests/hipemu/hip/hip_runtime.h (test infrastructure only)."""
import os, subprocess, sys, glob, hashlib, fcntl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, 'segtran_amd', 'csrc')
OUT = os.path.join(HERE, 'build')
CXX = '/opt/rocm/lib/llvm/bin/clang++'


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, '*.hip')))
    deps = srcs + glob.glob(os.path.join(SRC, '*.h')) + [os.path.join(HERE, 'hip', 'hip_runtime.h'),
                                                         os.path.join(ROOT, 'include', 'segx.h')]
    h = hashlib.sha1()
    for d in deps:
        h.update(open(d, 'rb').read())
    stamp = os.path.join(OUT, 'stamp')
    lib = os.path.join(OUT, 'libsegx_emu.so')
    # one builder at a time (pytest-xdist workers start together): the others wait on the lock and find the finished library
    with open(os.path.join(OUT, 'lock'), 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build_locked(srcs, h.hexdigest(), stamp, lib, verbose)


def _build_locked(srcs, digest, stamp, lib, verbose):
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == digest:
        return lib
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        cmd = [CXX, '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-Wno-unused-value', '-Wno-builtin-macro-redefined', '-Wno-psabi',
               '-I', HERE, '-c', s, '-o', o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError('hipemu build failed for %s:\n%s' % (s, out))
        if verbose and out:
            print(out)
    subprocess.check_call([CXX, '-shared', '-o', lib + '.tmp'] + objs)
    os.replace(lib + '.tmp', lib)                          # a reader never sees a half-written library
    open(stamp, 'w').write(digest)
    return lib


if __name__ == '__main__':
    print(build(verbose=True))
