#!/bin/bash
# full kernel names (torch's own element-wise launches included) of one model's training steps: bash tools/trace_train_names.sh HAT:4
set -eo pipefail
SPEC=${1:-HAT:4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_trace" -o t -- python3 "$ROOT/tools/train_bench.py" $SPEC > "$OUT/train_trace.log" 2>&1
cd "$ROOT"
python3 - "$OUT/train_trace" > "$OUT/train_names_${SPEC%%:*}.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::native" not in n and "rocclr" not in n:
            continue
        a = acc[n[:260]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t/4/1e3:7.2f} ms/step n/step={c/4:7.1f} avg {t/c:6.1f} us  {n}")
PY
rm -rf "$OUT/train_trace"
cat "$OUT/train_names_${SPEC%%:*}.txt"
