// sr_conv3x3 variant: compute bf16, input float, 8-row tiles (see sr_conv_impl.h)
#include "sr_conv_impl.h"
SR_CONV_VARIANT(sr_conv_dispatch_bf16_f32_8, bf16, float, 8)
