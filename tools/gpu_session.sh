#!/bin/bash
# One measurement session (run through gpurun from the repo root): tools/gpu_session.sh <tag> [tests|notests] [pmc] [extra command ...]
# full -m gpu suite, smoke, the driver's bench line (+ bench_shapes.json), rocprofv3 kernel stats + per-dispatch traces of cfg2 / cfg4 / cfg5, optional counter passes.
set -u
TAG=${1:-r05_x}; MODE=${2:-tests}; PMCMODE=${3:-nopmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx" > $OUT/box.txt
if [ "$MODE" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
if [ $# -gt 3 ]; then shift 3; for CMD in "$@"; do echo "== $CMD" >> $OUT/extra.log; timeout 600 bash -c "$CMD" >> $OUT/extra.log 2>&1; done; tail -5 $OUT/extra.log; fi
SEGX_BENCH_VERBOSE=2 timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_shapes.txt; wc -c $OUT/bench_default.json; cut -c1-400 $OUT/bench_default.json
cp bench_shapes.json $OUT/bench_shapes.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg4 cfg5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$CFG -o $CFG -- python $ROOT/bench.py --config $CFG --steps 2 --warmup 2 --no-brats --no-cpu-baseline --single-order > $OUT/prof_$CFG.log 2>&1
  find $OUT/prof_$CFG -name '*kernel_stats.csv' -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
  find $OUT/prof_$CFG -name '*kernel_trace.csv' -exec cp {} $OUT/${CFG}_kernel_trace.csv \;
  find $OUT/prof_$CFG -name '*agent_info.csv' -exec cp {} $OUT/agent_info.csv \;
  rm -rf $OUT/prof_$CFG
done
if [ "$PMCMODE" = pmc ]; then
  # counter passes (own runs, --pmc only: no trace domains beside them), aggregated per kernel on the box
  for CFG in cfg2 cfg4 cfg5; do
    for PMC in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
      NAME=$(echo $PMC | cut -d' ' -f1)
      timeout 400 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_${CFG}_$NAME -o pmc -- python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-brats --no-cpu-baseline --single-order > $OUT/pmc_${CFG}_$NAME.log 2>&1
      python $ROOT/tools/pmc_summary.py $OUT/pmc_${CFG}_$NAME $OUT/pmc_${CFG}_${NAME}_by_kernel.json > /dev/null 2>&1
      rm -rf $OUT/pmc_${CFG}_$NAME
    done
  done
fi
ls -la $OUT
