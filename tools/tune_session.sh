#!/bin/bash
# Device sweep for csrc/gemm_tuned.h (run through gpurun from the repo root): tools/tune_session.sh <tag> <shapes.txt> [<shapes.txt> ...]
# prints the GEMM shapes of cfg1 first (the other configurations' lists come from a bench session: SEGX_BENCH_VERBOSE=2), then times every tile x split-K candidate per shape.
set -u
TAG=${1:-r04_tune}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
SEGX_BENCH_VERBOSE=2 timeout 300 python bench.py --config cfg1 --no-brats --no-cpu-baseline --single-order > $OUT/bench_cfg1.json 2> $OUT/cfg1_shapes.txt
timeout 1500 python tools/tune_gemm.py "$@" $OUT/cfg1_shapes.txt > $OUT/tune_gemm.txt 2> $OUT/tune_gemm.err
tail -3 $OUT/tune_gemm.err; wc -l $OUT/tune_gemm.txt
