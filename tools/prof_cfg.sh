#!/bin/bash
# rocprofv3 kernel statistics of one or more bench configurations: tools/prof_cfg.sh <tag> cfg4 cfg5 ...   (run on the GPU box, from the repo root)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in "$@"; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$CFG -o $CFG -- python $ROOT/bench.py --config $CFG --steps 2 --warmup 2 --no-brats --no-cpu-baseline --single-order > $OUT/prof_$CFG.log 2>&1
  find $OUT/prof_$CFG -name '*kernel_stats.csv' -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
  rm -rf $OUT/prof_$CFG
done
ls $OUT
