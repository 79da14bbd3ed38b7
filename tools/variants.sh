#!/bin/bash
# Try library variants (studiosr_amd/lib/variants/*.so) on the GPU box: tools/variants.sh "<command>" -> one log per variant.
set -eo pipefail
CMD=${1:-python tools/kbench.py blk}
mkdir -p gpurun_out
for so in studiosr_amd/lib/variants/*.so; do
  name=$(basename "$so" .so)
  cp "$so" studiosr_amd/lib/libstudiosr_hip.so
  echo "== $name" | tee -a gpurun_out/variants.log
  timeout -k 10 300 $CMD >> gpurun_out/variants.log 2>&1
done
