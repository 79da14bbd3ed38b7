#!/bin/bash
# Collect the per-round rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_bench.sh r01'
# Writes gpurun_out/<tag>_*; copy the summaries you want judged into profiles/.
# Kernel trace and the two PMC passes are separate runs (gfx950: one HBM counter per pass, never together with tracing).
set -eo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -o bench -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/${TAG}_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace1" -o bench -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --inflight 1 --skip-cpu > "$OUT/${TAG}_trace1.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmc_fetch" -o bench -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --inflight 1 > "$OUT/${TAG}_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmc_write" -o bench -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --inflight 1 > "$OUT/${TAG}_pmc_write.log" 2>&1
cd "$ROOT"
python3 tools/trace_summary.py "$OUT/${TAG}_trace" > "$OUT/${TAG}_bench_kernel_trace_summary.txt"
python3 tools/trace_summary.py "$OUT/${TAG}_trace1" > "$OUT/${TAG}_bench_inflight1_kernel_trace_summary.txt"
# first line: what the dominant kernel's source looked like when the counters were taken (bench.py refuses a profile whose hash differs from the tree's)
{ echo "# kernel_src_sha16 = $(python3 -c 'import bench; print(bench.dominant_kernel_src_sha16())')"; python3 tools/pmc_summary.py "$OUT/${TAG}_pmc_fetch"; python3 tools/pmc_summary.py "$OUT/${TAG}_pmc_write"; } > "$OUT/${TAG}_bench_hbm_counters.txt"
find "$OUT/${TAG}_trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/${TAG}_bench_kernel_stats.csv" \;
# the raw traces are large; keep only the summaries
rm -rf "$OUT/${TAG}_trace" "$OUT/${TAG}_trace1" "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write"
cat "$OUT/${TAG}_bench.json"
