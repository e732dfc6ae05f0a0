#!/bin/bash
# Same-box A/B of one environment switch on a bench configuration: tools/ab_env.sh <cfg> <VAR=a> <VAR=b> [steps]   (alternating runs, 2 rounds)
CFG=${1:-cfg4}; A=${2}; B=${3}; STEPS=${4:-15}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for R in 1 2; do for KV in $A $B; do
  echo -n "$CFG $KV: "
  env $KV python bench.py --config $CFG --steps $STEPS --warmup 5 --no-brats --no-cpu-baseline --single-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
done; done
