#!/bin/bash
# Same-box A/B of one segx_tune knob on a bench configuration: tools/ab_knob.sh <cfg> <knob:valueA> <knob:valueB> [steps]   (alternating runs, 2 rounds)
CFG=${1:-cfg4}; A=${2:-16:0}; B=${3:-16:1}; STEPS=${4:-15}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for R in 1 2; do for KV in $A $B; do
  echo -n "$CFG SEGX_TUNE=$KV: "
  SEGX_TUNE=$KV python bench.py --config $CFG --steps $STEPS --warmup 5 --no-brats --no-cpu-baseline --single-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
done; done
