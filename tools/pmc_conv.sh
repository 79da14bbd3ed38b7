#!/bin/bash
# TA / SQ counters of the conv kernels at the bench shapes (gpurun -- 'bash tools/pmc_conv.sh'): separate --pmc passes.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_conv.txt
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | cut -c1-10 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmcc_$tag" -o k -- python3 "$ROOT/tools/kbench.py" conv > "$OUT/pmcc_$tag.log" 2>&1 || echo "set failed: $set" >> $OUT/pmc_conv.txt
  python3 $ROOT/tools/pmc_summary.py "$OUT/pmcc_$tag" >> $OUT/pmc_conv.txt
  rm -rf "$OUT/pmcc_$tag"
done
cat $OUT/pmc_conv.txt
