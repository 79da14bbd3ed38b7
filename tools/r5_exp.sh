#!/bin/bash
# round-5 block-kernel experiments: tools/r5_exp.sh "<Bs>" <variant>[:SR_BLOCK_WGS] ... ('shipped' = the in-tree library); log -> gpurun_out/r5_exp.log
set -o pipefail
mkdir -p gpurun_out
BS=$1; shift
for spec in "$@"; do
  name=${spec%%:*}; wgs=""; [ "$spec" != "$name" ] && wgs=${spec#*:}
  echo "== $spec" | tee -a gpurun_out/r5_exp.log
  if [ "$name" = shipped ]; then unset SR_LIB_PATH; else export SR_LIB_PATH="$PWD/studiosr_amd/lib/variants/$name.so"; fi
  SR_BLOCK_WGS=${wgs:-0} SR_BS=$BS timeout -k 10 200 python tools/blk3_time.py v3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5_exp.log
done
