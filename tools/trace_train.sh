#!/bin/bash
# rocprofv3 kernel trace of one model's training steps: bash tools/trace_train.sh HAT:4  -> gpurun_out/train_trace_<KIND>.txt
set -eo pipefail
SPEC=${1:-HAT:4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TRAIN_STEPS=${TRAIN_STEPS:-12}  # one-time work (parameter upload, flat-buffer setup, optimizer state) is amortised over the steps
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -o t -- python3 "$ROOT/tools/train_bench.py" $SPEC > "$OUT/train_trace.log" 2>&1
cd "$ROOT"
python3 - "$OUT/train_trace" > "$OUT/train_trace_${SPEC%%:*}.txt" <<'PY'
import csv, glob, sys, os
N = int(os.environ.get("TRAIN_STEPS", "12"))
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"sr_\w+(<[^>]*>)?", r["Kernel_Name"])
        n = m.group(0) if m else r["Kernel_Name"][:60]
        a = acc[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
print(f"total kernel time {tot/1e3:.1f} ms over {N} steps = {tot/1e3/N:.2f} ms/step, {sum(v[0] for v in acc.values())/N:.0f} launches/step (incl. one-time setup / {N})")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/N/1e3:8.2f} ms/step {100*t/tot:5.1f}%  n/step={c/N:7.1f}  avg {t/c:8.1f} us  {n}")
PY
rm -rf "$OUT/train_trace"
cat "$OUT/train_trace_${SPEC%%:*}.txt"
