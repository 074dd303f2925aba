// In-place mixed-radix kernels with compile-time stage lists, group A of generic_static_inplace.hpp (generic_static_inplace_impl.hpp).
#define CFFT_SIP_GROUP CFFT_STATIC_INPLACE_A
#define CFFT_SIP_ENTRY launch_static_inplace_a
#include "generic_static_inplace_impl.hpp"
