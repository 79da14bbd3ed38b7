#!/bin/bash
# Build a variant library with extra flags on every conv translation unit: tools/exp_conv.sh NAME -DSR_CONV_RING=5 ...  -> studiosr_amd/lib/variants/NAME.so
set -eo pipefail
cd "$(dirname "$0")/../studiosr_amd/csrc"
mkdir -p ../lib/variants
NAME=$1; shift
SRCS="sr_conv sr_conv_v_bf16_f32_4 sr_conv_v_bf16_bf16_4 sr_conv_v_bf16_f32_8 sr_conv_v_bf16_bf16_8 sr_conv_v_bf3_f32_4 sr_conv_v_f32_f32_4 sr_conv_big"
OBJS=$(ls ../lib/obj/*.o)
NEW=""
for s in $SRCS; do
    OBJS=$(echo "$OBJS" | grep -v "/$s.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $s.hip -o /tmp/${s}_$NAME.o &
    NEW="$NEW /tmp/${s}_$NAME.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$NAME.so $OBJS $NEW
echo built $NAME
