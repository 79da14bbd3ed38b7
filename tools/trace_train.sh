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
import re
acc = defaultdict(lambda: [0, 0.0])
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# One-time setup (parameter upload, flat-buffer construction: ~2,900 __amd_rocclr_copyBuffer launches -- round 4's "239 copyBuffer per step" were these divided by
# the step count) ends where the first step's sr_tr_gather starts; the first step (plan recording, lazy buffers) is dropped too.
gathers = [i for i, r in enumerate(rows) if "sr_tr_gather" in r["Kernel_Name"]]
setup = gathers[0] if gathers else 0
first = gathers[2] if len(gathers) > 2 else setup  # two gather launches per step
n_setup = setup
rows = rows[first:]
N = max(1, N - 1)
busy, end = 0.0, 0
for r in rows:
    m = re.search(r"sr_\w+(<[^>]*>)?", r["Kernel_Name"])
    n = m.group(0) if m else r["Kernel_Name"][:60]
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = acc[n]; a[0] += 1; a[1] += (e_ - s_) / 1e3
    if e_ > end:
        busy += (e_ - max(s_, end)) / 1e3
        end = e_
tot = sum(v[1] for v in acc.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 if rows else 0.0
print(f"{n_setup} launches of one-time setup dropped; steps 2..{N + 1}: total kernel time {tot/1e3:.1f} ms = {tot/1e3/N:.2f} ms/step, {sum(v[0] for v in acc.values())/N:.0f} launches/step; "
      f"GPU busy (union of kernel intervals) {busy/1e3/N:.2f} ms/step of {span/1e3/N:.2f} ms/step wall")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/N/1e3:8.2f} ms/step {100*t/tot:5.1f}%  n/step={c/N:7.1f}  avg {t/c:8.1f} us  {n}")
PY
rm -rf "$OUT/train_trace"
cat "$OUT/train_trace_${SPEC%%:*}.txt"
