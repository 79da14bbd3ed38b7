#!/bin/bash
# HBM counters (FETCH_SIZE, WRITE_SIZE; separate passes, no tracing) of one model's eval forward: bash tools/pmc_model.sh HAT:4
set -eo pipefail
SPEC=${1:-HAT:4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
K=${SPEC%%:*}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f_$K" -o m -- python3 "$ROOT/tools/model_bench.py" $SPEC > "$OUT/pmc_model.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w_$K" -o m -- python3 "$ROOT/tools/model_bench.py" $SPEC >> "$OUT/pmc_model.log" 2>&1
cd "$ROOT"
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB per launch, raw: double FETCH_SIZE on gfx950) -- python3 tools/model_bench.py $SPEC"; python3 tools/pmc_summary.py "$OUT/pmc_f_$K"; python3 tools/pmc_summary.py "$OUT/pmc_w_$K"; } > "$OUT/model_hbm_counters_$K.txt"
rm -rf "$OUT/pmc_f_$K" "$OUT/pmc_w_$K"
head -60 "$OUT/model_hbm_counters_$K.txt"
