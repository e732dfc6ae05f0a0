#!/bin/bash
# counter passes over tools/conv_one.py: tools/pmc_one.sh <tag> <args of conv_one.py ...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PMC in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  NAME=$(echo $PMC | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p_$NAME -o pmc -- python $ROOT/tools/conv_one.py "$@" > $OUT/p_$NAME.log 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/p_$NAME $OUT/${NAME}_by_kernel.json > /dev/null 2>&1
  rm -rf $OUT/p_$NAME
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python $ROOT/tools/conv_one.py "$@" > $OUT/ks.log 2>&1
find $OUT/ks -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \; ; rm -rf $OUT/ks
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*_by_kernel.json')):
    d=json.load(open(f))
    for k,v in d.items():
        if 'halo' in k and 'pack' not in k and 'reduce' not in k: print(os.path.basename(f)[:30], k[:60], v)
PY
grep halo $OUT/kernel_stats.csv | cut -c1-200
