// sr_conv3x3 variant: compute bf16, input bf16, 8-row tiles (see sr_conv_impl.h)
#include "sr_conv_impl.h"
SR_CONV_VARIANT(sr_conv_dispatch_bf16_bf16_8, bf16, bf16, 8)

// in-kernel stamps of THIS variant (the bf16 8-row tiles; `make STAMPS=1`, tools/kbench.py conv with KB_STAMPS=1)
extern "C" int sr_debug_conv_stamps(unsigned long long* host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(sr_conv_impl::sr_dbg_conv), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
