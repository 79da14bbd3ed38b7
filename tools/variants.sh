#!/bin/bash
# Try library variants (studiosr_amd/lib/variants/*.so) on the GPU box: tools/variants.sh "<command>" -> one log per variant.
# The variant is selected through SR_LIB_PATH (read by studiosr_amd/_lib.py); the shipped library is never overwritten.
set -eo pipefail
CMD=${1:-python tools/kbench.py blk}
mkdir -p gpurun_out
for so in studiosr_amd/lib/variants/*.so; do
  name=$(basename "$so" .so)
  echo "== $name" | tee -a gpurun_out/variants.log
  SR_LIB_PATH="$PWD/$so" timeout -k 10 300 $CMD >> gpurun_out/variants.log 2>&1
done
