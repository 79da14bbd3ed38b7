#!/bin/bash
# rocprofv3 kernel trace of an arbitrary tools/*.py script -> gpurun_out/trace_script.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/ts" -o t -- python3 "$ROOT/$SCRIPT" "$@" > "$OUT/ts.log" 2>&1
cd "$ROOT"
python3 tools/trace_summary.py "$OUT/ts" > "$OUT/trace_script.txt"
rm -rf "$OUT/ts"
