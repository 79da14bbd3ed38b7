// sr_conv3x3 variant: compute bf16, input float, 4-row tiles (see sr_conv_impl.h)
#include "sr_conv_impl.h"
SR_CONV_VARIANT(sr_conv_dispatch_bf16_f32_4, bf16, float, 4)
