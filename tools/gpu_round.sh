#!/bin/bash
# One measurement session on the MI355X box (run through gpurun from the repo root):
#   tools/gpu_round.sh <tag> [full|quick|bench|micro]
# Writes everything under gpurun_out/<tag>/ ; copy what should be judged into profiles/ afterwards (tools/collect_profiles.py).
set -u
TAG=${1:-r02_x}; MODE=${2:-full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx" > $OUT/box.txt
if [ "$MODE" = full ] || [ "$MODE" = quick ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
if [ "$MODE" != micro ]; then
  # the driver's command (main cfg2 + brats cfg4/cfg5 blocks + cpu baseline), bf16x6 engine (default)
  SEGX_BENCH_VERBOSE=2 timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_shapes.txt; cut -c1-300 $OUT/bench_default.json
  # the fp32-MFMA engine on the same workloads (comparison; no cpu baseline)
  for CFG in cfg2 cfg4; do
    SEGX_BENCH_VERBOSE=2 timeout 600 python bench.py --config $CFG --engine f32 --steps 12 --warmup 4 --no-brats --no-cpu-baseline --single-order > $OUT/bench_${CFG}_f32.json 2> $OUT/bench_${CFG}_f32_shapes.txt
    cut -c1-200 $OUT/bench_${CFG}_f32.json
  done
fi
if [ "$MODE" != micro ]; then
  # the launch-bound configuration eagerly and as one hipGraph per step
  for G in "" "--graph"; do
    timeout 600 python bench.py --config cfg1 $G --no-brats --no-cpu-baseline --single-order > $OUT/bench_cfg1${G}.json 2> $OUT/bench_cfg1${G}.err; cut -c1-200 $OUT/bench_cfg1${G}.json
  done
fi
if [ "$MODE" = full ] || [ "$MODE" = micro ]; then
  timeout 600 python tools/gemm_bench.py 8 tiles > $OUT/gemm_bench_tiles.txt 2>&1; tail -5 $OUT/gemm_bench_tiles.txt
fi
if [ "$MODE" = bench ] || [ "$MODE" = micro ]; then ls -la $OUT; exit 0; fi
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg4 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$CFG -o $CFG -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 2 --no-brats --no-cpu-baseline --single-order > $OUT/prof_$CFG.log 2>&1
  find $OUT/prof_$CFG -name '*kernel_stats.csv' -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
  find $OUT/prof_$CFG -name '*agent_info.csv' -exec cp {} $OUT/agent_info.csv \;
  rm -rf $OUT/prof_$CFG
done
if [ "$MODE" = full ]; then
  for CFG in cfg2 cfg4 cfg5; do
    for PMC in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
      NAME=$(echo $PMC | cut -d' ' -f1)
      timeout 600 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_${CFG}_$NAME -o pmc -- python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-brats --no-cpu-baseline --single-order > $OUT/pmc_${CFG}_$NAME.log 2>&1
      python $ROOT/tools/pmc_summary.py $OUT/pmc_${CFG}_$NAME $OUT/pmc_${CFG}_${NAME}_by_kernel.json > /dev/null 2>&1
      rm -rf $OUT/pmc_${CFG}_$NAME
    done
  done
fi
ls -la $OUT
