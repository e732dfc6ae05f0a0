set -u
OUT=gpurun_out/r02_c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_conv3d.py tests/test_kernels_gemm.py tests/test_gpu_model.py tests/test_gpu_fullshape.py -m gpu -q -k "conv3d or bridge or x6 or split_early or 3d or cfg4 or cfg5" > $OUT/pytest_3d.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_3d.log; tail -4 $OUT/pytest_3d.log
python - > $OUT/variants67.txt 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.argv=['gemm_bench.py','8','none']
exec(open('tools/gemm_bench.py').read().split("if len(sys.argv) > 2 and sys.argv[2] == 'variants':")[0])
VN = {0: 'product', 6: 'split-early 2w', 7: 'product 2w'}
for sh in (MAIN[0], MAIN[8], MAIN[9], MAIN[6], MAIN[7]):
    for v in (0, 6, 7, 0, 6):
        L.c.segx_tune(6, v)
        run('%s [%s]' % (sh[0][:14], VN[v]), *sh[1:7], nb=sh[7], engine='x6', tile=1)
    L.c.segx_tune(6, 0)
PY
tail -12 $OUT/variants67.txt
for CFG in cfg4 cfg5; do
  SEGX_BENCH_VERBOSE=2 timeout 600 python bench.py --config $CFG --steps 16 --warmup 5 --no-brats --no-cpu-baseline > $OUT/bench_$CFG.json 2> $OUT/bench_${CFG}_shapes.txt; cut -c1-220 $OUT/bench_$CFG.json
done
