"""Bench-only builds of libsegx with extra compiler flags on ONE source (A/B of kernel variants on the GPU box, tools/ws_bench.py):
python tools/build_variant.py <name> <source.hip> <flag> [<flag> ...]  ->  tools/variants/libsegx_<name>.so  (never loaded by the product)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segtran_amd import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()                                                                     # the product objects (segtran_amd/lib/*.o)
out = os.path.join(ROOT, 'tools', 'variants')
os.makedirs(out, exist_ok=True)
srcs = src.split(',')                                                         # several sources: gemm.hip,fpn.hip (segx_tune lives in fpn.hip: -DSEGX_BENCH must reach it too)
objs = []
for one in srcs:
    obj = os.path.join(out, '%s_%s.o' % (os.path.basename(one), name))
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + flags + ['-c', os.path.join(B.SRC, one), '-o', obj])
    objs.append(obj)
others = [o for o in sorted(glob.glob(os.path.join(B.OUT, '*.hip.o'))) if os.path.basename(o)[:-2] not in srcs]
lib = os.path.join(out, 'libsegx_%s.so' % name)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + others)
print(lib)
