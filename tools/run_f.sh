set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_f; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
SEGX_BENCH_VERBOSE=2 timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_shapes.txt; cut -c1-300 $OUT/bench_default.json
for CFG in cfg1; do
  for G in "" "--graph"; do
    timeout 600 python bench.py --config $CFG $G --no-brats --no-cpu-baseline --single-order > $OUT/bench_${CFG}${G}.json 2> $OUT/bench_${CFG}${G}.err; cut -c1-200 $OUT/bench_${CFG}${G}.json; tail -3 $OUT/bench_${CFG}${G}.err
  done
done
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg4 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$CFG -o $CFG -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 2 --no-brats --no-cpu-baseline --single-order > $OUT/prof_$CFG.log 2>&1
  find $OUT/prof_$CFG -name '*kernel_stats.csv' -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
  rm -rf $OUT/prof_$CFG
done
ls -la $OUT
