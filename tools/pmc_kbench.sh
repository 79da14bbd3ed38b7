#!/bin/bash
# SQ counter passes over tools/kbench.py (gpurun -- 'bash tools/pmc_kbench.sh blk').  Two passes of <= 8 SQ counters each.
set -eo pipefail
WHICH=${1:-blk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_a" -o k -- python3 "$ROOT/tools/kbench.py" $WHICH > "$OUT/pmc_a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_b" -o k -- python3 "$ROOT/tools/kbench.py" $WHICH > "$OUT/pmc_b.log" 2>&1
cd "$ROOT"
{ python3 tools/pmc_summary.py "$OUT/pmc_a"; python3 tools/pmc_summary.py "$OUT/pmc_b"; } > "$OUT/pmc_kbench_$WHICH.txt"
rm -rf "$OUT/pmc_a" "$OUT/pmc_b"
