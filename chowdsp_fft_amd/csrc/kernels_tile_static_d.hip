// Tile kernels with compile-time stage lists, group D of tables.hpp's lengths (tile_static_impl.hpp).
#define CFFT_TILE_GROUP CFFT_STATIC_TILE_SIZES_D
#define CFFT_TILE_PICK_A pick_static_pass_a_d
#define CFFT_TILE_PICK_B pick_static_pass_b_d
#define CFFT_TILE_PICK_COL pick_static_column_d
#include "tile_static_impl.hpp"
