#!/bin/bash
# Which launches surround the __amd_rocclr_copyBuffer kernels of a fused HAT training step?  bash tools/trace_copybuffer.sh -> gpurun_out/train_copybuffer.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TRAIN_STEPS=4
rocprofv3 --kernel-trace --output-format csv -d "$OUT/cb_trace" -o t -- python3 "$ROOT/tools/train_bench.py" HAT:4 > "$OUT/cb_trace.log" 2>&1
cd "$ROOT"
python3 - "$OUT/cb_trace" > "$OUT/train_copybuffer.txt" <<'PY'
import csv, glob, sys, re, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    m = re.search(r"sr_\w+", r["Kernel_Name"]); return m.group(0) if m else r["Kernel_Name"][:40]
print("columns:", list(rows[0].keys()))
ctx = collections.Counter(); sizes = collections.Counter()
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = nm(rows[i - 1]) if i else "-"; nxt = nm(rows[i + 1]) if i + 1 < len(rows) else "-"
        ctx[(prev, nxt, r.get("Queue_Id"), r.get("Stream_Id"))] += 1
        sizes[(r.get("Grid_Size"), r.get("Workgroup_Size"))] += 1
print("copyBuffer launches:", sum(ctx.values()))
for k, c in ctx.most_common(30): print(c, k)
print("grid sizes:")
for k, c in sizes.most_common(10): print(c, k)
PY
rm -rf "$OUT/cb_trace"
cat "$OUT/train_copybuffer.txt"
