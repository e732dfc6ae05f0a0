#!/bin/bash
# A short device session: selected tests, micro-benchmarks, quick bench lines.  tools/gpu_quick.sh <tag> "<pytest -k expression>" [extra commands...]
set -u
TAG=${1:-r05_q}; KEXPR=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -k "$KEXPR" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log; tail -5 $OUT/pytest_sel.log
fi
shift 2
i=0
for CMD in "$@"; do i=$((i+1)); echo "== $CMD" > $OUT/cmd$i.log; timeout 600 bash -c "$CMD" >> $OUT/cmd$i.log 2>&1; echo "rc=$?" >> $OUT/cmd$i.log; tail -3 $OUT/cmd$i.log; done
ls -la $OUT
