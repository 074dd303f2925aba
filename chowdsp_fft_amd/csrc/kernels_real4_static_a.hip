// The real four-step's tile kernels with compile-time stage lists, group A of tables.hpp's lengths (real4_static_impl.hpp).
#define CFFT_R4S_GROUP CFFT_STATIC_TILE_SIZES_A
#define CFFT_R4S_SUFFIX a
#include "real4_static_impl.hpp"
