#!/bin/bash
# One measurement session on the MI355X box (run through gpurun from the repo root):
#   tools/gpu_round.sh <tag> [full|quick]
# Writes everything under gpurun_out/<tag>/ ; copy what should be judged into profiles/ afterwards (tools/collect_profiles.py).
set -u
TAG=${1:-r01_x}; MODE=${2:-full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$MODE" = prof ]; then SKIP_BENCH=1; MODE=full; SKIP_TESTS=1; fi
if [ "$MODE" = full ] && [ -z "${SKIP_TESTS:-}" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
for CFG in cfg2 cfg4; do
  [ -n "${SKIP_BENCH:-}" ] && continue
  SEGX_BENCH_VERBOSE=2 timeout 600 python bench.py --config $CFG --steps 8 --warmup 3 > $OUT/bench_$CFG.json 2> $OUT/bench_${CFG}_gemm_shapes.txt
  cut -c1-260 $OUT/bench_$CFG.json
done
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$CFG -o $CFG -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --single-order > $OUT/prof_$CFG.log 2>&1
done
if [ "$MODE" = full ]; then
  for CFG in cfg2 cfg4; do
    for PMC in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_${CFG}_$PMC -o pmc -- python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --single-order > $OUT/pmc_${CFG}_$PMC.log 2>&1
      python $ROOT/tools/pmc_summary.py $OUT/pmc_${CFG}_$PMC $OUT/pmc_${CFG}_${PMC}_by_kernel.json > /dev/null 2>&1
      rm -rf $OUT/pmc_${CFG}_$PMC
    done
  done
fi
# keep only the summaries (the raw traces are tens of MB)
for CFG in cfg2 cfg4; do
  find $OUT/prof_$CFG -name '*kernel_stats.csv' -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
  find $OUT/prof_$CFG -name '*agent_info.csv' -exec cp {} $OUT/agent_info.csv \;
  rm -rf $OUT/prof_$CFG
done
ls -la $OUT
