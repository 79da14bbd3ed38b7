#!/bin/bash
# rocprofv3 kernel trace of one model's eval forward (tools/model_bench.py): bash tools/trace_model.sh HAT:4 -> gpurun_out/model_trace_<KIND>.txt
set -eo pipefail
SPEC=${1:-HAT:4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/model_trace" -o t -- python3 "$ROOT/tools/model_bench.py" $SPEC > "$OUT/model_trace.log" 2>&1
cd "$ROOT"
python3 - "$OUT/model_trace" > "$OUT/model_trace_${SPEC%%:*}.txt" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = defaultdict(lambda: [0, 0.0])
for r in rows:
    m = re.search(r"sr_\w+(<[^>]*>)?", r["Kernel_Name"])
    n = (m.group(0) if m else r["Kernel_Name"][:60]) + f" grid={int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))}x{r['Grid_Size_Y']}"
    a = acc[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
print(f"total kernel time {tot/1e3:.1f} ms, {len(rows)} launches (all replays + warm-up)")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{100*t/tot:5.1f}%  n={c:6d}  avg {t/c:8.1f} us  {n}")
PY
rm -rf "$OUT/model_trace"
cat "$OUT/model_trace_${SPEC%%:*}.txt"
