#!/bin/bash
# GEMM-engine session on the MI355X box: tools/gpu_ws.sh <tag>  (A/B of the bf16x6 kernels + matrix-pipe counters)
set -u
TAG=${1:-r03_a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd $ROOT
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx" > $OUT/box.txt
timeout 900 python -m pytest tests/test_kernels_gemm.py -m gpu -q -k "x6" > $OUT/pytest_x6.log 2>&1; tail -3 $OUT/pytest_x6.log
timeout 900 python tools/ws_bench.py 5 6 main > $OUT/ws_bench_main.txt 2>&1; grep -c TF $OUT/ws_bench_main.txt
timeout 400 python tools/ws_bench.py 3 6 backbone > $OUT/ws_bench_backbone.txt 2>&1
timeout 400 python tools/ws_bench.py 3 6 ablate > $OUT/ws_bench_ablate.txt 2>&1; grep -v '^/opt' $OUT/ws_bench_ablate.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ws -o pmc -- python $ROOT/tools/ws_bench.py 1 1 pmc > $OUT/pmc_ws.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/pmc_ws $OUT/pmc_ws_by_kernel.json > /dev/null 2>&1
rm -rf $OUT/pmc_ws
ls -la $OUT
