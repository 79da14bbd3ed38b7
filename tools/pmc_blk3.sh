#!/bin/bash
# SQ / TA / TCP counter passes over the round-3 Swin block kernel (gpurun -- 'bash tools/pmc_blk3.sh'): separate --pmc runs, <= 8 SQ counters each.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_blk3.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc3_$tag" -o k -- python3 "$ROOT/tools/blk3_time.py" v3 > "$OUT/pmc3_$tag.log" 2>&1 || echo "set failed: $set" >> $OUT/pmc_blk3.txt
  python3 $ROOT/tools/pmc_summary.py "$OUT/pmc3_$tag" >> $OUT/pmc_blk3.txt
  rm -rf "$OUT/pmc3_$tag"
done
cat $OUT/pmc_blk3.txt
