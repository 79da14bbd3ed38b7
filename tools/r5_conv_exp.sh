#!/bin/bash
# round-5 wide-conv experiments: short tiles / K phases / workgroups per CU for the 192 -> 192 conv (variants built with SR_EXP_SRC=sr_conv_big tools/exp3.sh NAME -DSR_BIG_TH=.. -DSR_BIG_PH=.. -DSR_BIG_OCC=..)
set -o pipefail
mkdir -p gpurun_out
for name in shipped "$@" shipped; do
  echo "== $name" | tee -a gpurun_out/r5_conv_exp.log
  if [ "$name" = shipped ]; then unset SR_LIB_PATH; else export SR_LIB_PATH="$PWD/studiosr_amd/lib/variants/$name.so"; fi
  timeout -k 10 200 python tools/kbench.py conv 2>&1 | grep "192->192" | tee -a gpurun_out/r5_conv_exp.log
done
