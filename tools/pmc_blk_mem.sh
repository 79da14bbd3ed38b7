#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/pmc_list.txt 2>&1 || true
grep -oE "\b(TA_[A-Z_]+|TCP_[A-Z_]+|TD_[A-Z_]+|SQ_INSTS_[A-Z_]+|SQ_INST_CYCLES_[A-Z_]+|SQ_IFETCH[A-Z_]*|SQC_[A-Z_]+)\b" $OUT/pmc_list.txt | sort -u > $OUT/pmc_names.txt
wc -l $OUT/pmc_names.txt
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_x_$tag" -o k -- python3 "$ROOT/tools/kbench.py" blk > "$OUT/pmc_x_$tag.log" 2>&1 || echo "set failed: $set"
  python3 $ROOT/tools/pmc_summary.py "$OUT/pmc_x_$tag" | grep -A12 "', 648)" >> $OUT/pmc_x_summary.txt
  rm -rf "$OUT/pmc_x_$tag"
done
cat $OUT/pmc_x_summary.txt
