#!/bin/bash
# Build experiment variants of the round-3 Swin block kernel into studiosr_amd/lib/variants/<name>.so
# (tools/exp3.sh NAME "-DSR_STAMPS" ...); select one with SR_LIB_PATH=<path> (studiosr_amd/_lib.py).
set -eo pipefail
cd "$(dirname "$0")/../studiosr_amd/csrc"
mkdir -p ../lib/variants
NAME=$1; shift
SRC=${SR_EXP_SRC:-sr_swin_block3}  # SR_EXP_SRC=sr_swin_tail builds a variant of another translation unit
OBJS=$(ls ../lib/obj/*.o | grep -v $SRC.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c $SRC.hip -o /tmp/${SRC}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$NAME.so $OBJS /tmp/${SRC}_$NAME.o
echo built $NAME
