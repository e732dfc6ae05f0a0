#!/bin/bash
# tools/gpu_tune.sh <tag>: collect the GEMM shapes of cfg1 / cfg3 (cfg2 / cfg4 / cfg5 come from the committed bench_default_shapes), then time every
# tile / split-K candidate per shape (tools/tune_gemm.py)
set -u
TAG=${1:-r03_t}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd $ROOT
for CFG in cfg1 cfg3; do  [ -f profiles/r03_w_shapes_$CFG.txt ] && cp profiles/r03_w_shapes_$CFG.txt $OUT/shapes_$CFG.txt && continue
  SEGX_BENCH_VERBOSE=2 timeout 300 python bench.py --config $CFG --steps 3 --warmup 2 --no-brats --no-cpu-baseline --single-order > $OUT/bench_$CFG.json 2> $OUT/shapes_$CFG.txt
done
timeout 1500 python tools/tune_gemm.py profiles/r03_i_bench_default_shapes.txt $OUT/shapes_cfg1.txt $OUT/shapes_cfg3.txt > $OUT/tune_gemm.txt 2> $OUT/tune_gemm.err
tail -n 3 $OUT/tune_gemm.txt; tail -n 3 $OUT/tune_gemm.err
