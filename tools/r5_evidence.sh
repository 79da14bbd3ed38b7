#!/bin/bash
# Round-5 evidence in one box: bench + traces + stamped counters (tools/profile_bench.sh), held clock (wgtrace build), model tables, train lines.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash tools/profile_bench.sh r05 > gpurun_out/r05_profile_bench.log 2>&1 || echo "profile_bench failed"
cd "$ROOT"
# (studiosr_amd/lib/variants/wgtrace.so must be current: rebuild it with tools/exp3.sh wgtrace -DSR_WGTRACE after ANY change under csrc/ -- it links the other objects as they are)
SR_LIB_PATH="$ROOT/studiosr_amd/lib/variants/wgtrace.so" timeout -k 10 200 python tools/wgtrace_blk3.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r05_block_kernel_clock.txt
timeout -k 10 300 python tools/model_bench.py EDSR SwinIR RCAN HAT SwinIR-light HAT:16 HAT:1 RCAN:8 EDSR:8 SwinIR:16 SwinIR:32 2>/dev/null > gpurun_out/r05_model_bench.jsonl
MB_PREC=fp32x3 timeout -k 10 300 python tools/model_bench.py EDSR SwinIR RCAN RCAN:8 HAT HAT:16 HAT:1 SwinIR-light 2>/dev/null >> gpurun_out/r05_model_bench.jsonl
timeout -k 10 200 python bench.py --mode train 2>/dev/null | tail -1 > gpurun_out/r05_bench_train.json
timeout -k 10 300 python bench.py --mode train --model swinir --steps 10 2>/dev/null | tail -1 >> gpurun_out/r05_bench_train.json
bash tools/trace_train.sh HAT:4 > /dev/null 2>&1
bash tools/trace_train.sh SwinIR:4 > /dev/null 2>&1
bash tools/pmc_train.sh > /dev/null 2>&1
cd "$ROOT"
tail -2 gpurun_out/r05_bench.json | cut -c1-600; cat gpurun_out/r05_block_kernel_clock.txt | head -4; cut -c1-170 gpurun_out/r05_model_bench.jsonl; cut -c1-260 gpurun_out/r05_bench_train.json
