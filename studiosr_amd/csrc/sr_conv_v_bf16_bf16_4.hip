// sr_conv3x3 variant: compute bf16, input bf16, 4-row tiles (see sr_conv_impl.h)
#include "sr_conv_impl.h"
SR_CONV_VARIANT(sr_conv_dispatch_bf16_bf16_4, bf16, bf16, 4)
