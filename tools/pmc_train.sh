#!/bin/bash
# HBM counters (FETCH_SIZE, WRITE_SIZE; separate passes, no tracing) of the fused HAT training step: bash tools/pmc_train.sh  -> gpurun_out/train_hbm_counters_HAT.txt
# (KiB per launch, mean over the launches of each (kernel, grid); FETCH_SIZE raw: double it on gfx950 before comparing with a byte count, MI355X_MICROARCH.md "HBM")
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TRAIN_STEPS=4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_tf" -o m -- python3 "$ROOT/tools/train_bench.py" HAT:4 > "$OUT/pmc_train.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_tw" -o m -- python3 "$ROOT/tools/train_bench.py" HAT:4 >> "$OUT/pmc_train.log" 2>&1
cd "$ROOT"
SHA=$(cat studiosr_amd/csrc/sr_tr_block.hip studiosr_amd/csrc/sr_tr_wgrad.hip studiosr_amd/csrc/sr_tr_attn.hip studiosr_amd/csrc/sr_tr_attn_lds.hip | sha256sum | cut -c1-16)
{ echo "# kernel_src_sha16 = $SHA   (sha256 over csrc/sr_tr_block.hip sr_tr_wgrad.hip sr_tr_attn.hip sr_tr_attn_lds.hip)";
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (KiB per launch, raw: double FETCH_SIZE on gfx950) -- python3 tools/train_bench.py HAT:4, 4 steps";
  python3 tools/pmc_summary.py "$OUT/pmc_tf"; python3 tools/pmc_summary.py "$OUT/pmc_tw"; } > "$OUT/train_hbm_counters_HAT.txt"
rm -rf "$OUT/pmc_tf" "$OUT/pmc_tw"
grep -A2 "tr_wgrad\|tr_attn_bwd\|tr_tail_bwd\|tr_qkv_bwd\|tr_tail_fwd" "$OUT/train_hbm_counters_HAT.txt" | head -80
