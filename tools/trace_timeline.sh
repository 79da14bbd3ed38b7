#!/bin/bash
# Launch timeline (start offset, duration, queue) of a slice of one model's last graph replay: where the critical path of a block runs.
#   bash tools/trace_timeline.sh HAT:4 [first_kernel_index] [count]  -> gpurun_out/model_timeline_<KIND>.txt
set -eo pipefail
SPEC=${1:-HAT:4}
FIRST=${2:-40}
COUNT=${3:-60}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/model_tl" -o t -- python3 "$ROOT/tools/model_bench.py" $SPEC > "$OUT/model_tl.log" 2>&1
cd "$ROOT"
python3 - "$OUT/model_tl" "$FIRST" "$COUNT" > "$OUT/model_timeline_${SPEC%%:*}.txt" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay = from the last launch of the forward's first kernel (sr_ingest_*) on
cut = max((i for i, r in enumerate(rows) if "ingest" in r["Kernel_Name"]), default=0)
last = rows[cut:]
t0 = int(last[0]["Start_Timestamp"])
print(f"last replay: {len(last)} launches, {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
first, count = int(sys.argv[2]), int(sys.argv[3])
for r in last[first:first + count]:
    m = re.search(r"sr_\w+", r["Kernel_Name"])
    n = m.group(0) if m else r["Kernel_Name"][:40]
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f}  q{r['Queue_Id']:>3}  {n} grid={wg}x{r['Grid_Size_Y']}")
PY
rm -rf "$OUT/model_tl"
cat "$OUT/model_timeline_${SPEC%%:*}.txt"
