// Fused convolution of mixed-radix frames, in-place kernel with compile-time stage lists, group A (generic_static_inplace_conv_impl.hpp).
#define CFFT_SIP_GROUP CFFT_STATIC_INPLACE_A
#define CFFT_SIPC_ENTRY launch_static_inplace_conv_a
#include "generic_static_inplace_conv_impl.hpp"
