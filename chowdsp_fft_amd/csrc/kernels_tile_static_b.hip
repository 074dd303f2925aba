// Tile kernels with compile-time stage lists, group B of tables.hpp's lengths (tile_static_impl.hpp).
#define CFFT_TILE_GROUP CFFT_STATIC_TILE_SIZES_B
#define CFFT_TILE_PICK_A pick_static_pass_a_b
#define CFFT_TILE_PICK_B pick_static_pass_b_b
#define CFFT_TILE_PICK_COL pick_static_column_b
#include "tile_static_impl.hpp"
