#!/bin/bash
# rocprofv3 kernel trace of tools/kbench.py <args> -> gpurun_out/trace_kbench.txt   (KB_CONV=... selects one conv case)
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tk" -o t -- python3 "$ROOT/tools/kbench.py" "$@" > "$OUT/tk.log" 2>&1
cd "$ROOT"
python3 tools/trace_summary.py "$OUT/tk" > "$OUT/trace_kbench.txt"
rm -rf "$OUT/tk"
