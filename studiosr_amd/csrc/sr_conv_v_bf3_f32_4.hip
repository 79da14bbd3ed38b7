// sr_conv3x3 variant: compute bf3, input float, 4-row tiles (see sr_conv_impl.h)
#include "sr_conv_impl.h"
SR_CONV_VARIANT(sr_conv_dispatch_bf3_f32_4, bf3, float, 4)
