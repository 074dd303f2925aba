// Fused convolution of mixed-radix frames, in-place kernel with compile-time stage lists, group D (generic_static_inplace_conv_impl.hpp).
#define CFFT_SIP_GROUP CFFT_STATIC_INPLACE_D
#define CFFT_SIPC_ENTRY launch_static_inplace_conv_d
#include "generic_static_inplace_conv_impl.hpp"
