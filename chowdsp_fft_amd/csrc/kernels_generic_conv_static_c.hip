// Fused convolution of mixed-radix frames, in-place kernel with compile-time stage lists, group C (generic_static_inplace_conv_impl.hpp).
#define CFFT_SIP_GROUP CFFT_STATIC_INPLACE_C
#define CFFT_SIPC_ENTRY launch_static_inplace_conv_c
#include "generic_static_inplace_conv_impl.hpp"
