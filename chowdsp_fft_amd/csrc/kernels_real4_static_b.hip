// The real four-step's tile kernels with compile-time stage lists, group B of tables.hpp's lengths (real4_static_impl.hpp).
#define CFFT_R4S_GROUP CFFT_STATIC_TILE_SIZES_B
#define CFFT_R4S_SUFFIX b
#include "real4_static_impl.hpp"
