#!/bin/bash
# rocprofv3 kernel trace of tools/model_bench.py for the given models -> gpurun_out/trace_<models>.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$(echo "$@" | tr ' ' '_')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tm_$TAG" -o t -- python3 "$ROOT/tools/model_bench.py" "$@" > "$OUT/tm_$TAG.log" 2>&1
cd "$ROOT"
python3 tools/trace_summary.py "$OUT/tm_$TAG" > "$OUT/trace_$TAG.txt"
rm -rf "$OUT/tm_$TAG"
