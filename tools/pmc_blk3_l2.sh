#!/bin/bash
# L2 (TCC) hit / miss counters of the round-3 block kernel (gpurun -- 'bash tools/pmc_blk3_l2.sh')
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_blk3_l2.txt
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmcl2_$tag" -o k -- python3 "$ROOT/tools/blk3_time.py" v3 > "$OUT/pmcl2_$tag.log" 2>&1 || echo "set failed: $set" >> $OUT/pmc_blk3_l2.txt
  python3 $ROOT/tools/pmc_summary.py "$OUT/pmcl2_$tag" >> $OUT/pmc_blk3_l2.txt
  rm -rf "$OUT/pmcl2_$tag"
done
cat $OUT/pmc_blk3_l2.txt
