// sr_conv3x3 variant: compute float, input float, 4-row tiles (see sr_conv_impl.h)
#include "sr_conv_impl.h"
SR_CONV_VARIANT(sr_conv_dispatch_f32_f32_4, float, float, 4)
