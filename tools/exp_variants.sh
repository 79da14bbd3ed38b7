#!/bin/bash
# Build timing-only experiment variants of the Swin block kernel into studiosr_amd/lib/variants/<name>.so
# (tools/exp_variants.sh NAME "-DSR_EXP_W0" ...); run them with SR_LIB_PATH or tools/variants.sh.
set -eo pipefail
cd "$(dirname "$0")/../studiosr_amd/csrc"
mkdir -p ../lib/variants
NAME=$1; shift
OBJS=$(ls ../lib/obj/*.o | grep -v sr_swin_block.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c sr_swin_block.hip -o /tmp/sr_swin_block_$NAME.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$NAME.so $OBJS /tmp/sr_swin_block_$NAME.o
echo built $NAME
