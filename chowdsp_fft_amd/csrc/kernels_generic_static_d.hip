// In-place mixed-radix kernels with compile-time stage lists, group D of generic_static_inplace.hpp (generic_static_inplace_impl.hpp).
#define CFFT_SIP_GROUP CFFT_STATIC_INPLACE_D
#define CFFT_SIP_ENTRY launch_static_inplace_d
#include "generic_static_inplace_impl.hpp"
