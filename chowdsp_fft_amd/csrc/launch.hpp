// Kernel launchers (defined in kernels_*.hip) used by the C-ABI layer (capi.cpp).
#pragma once
#include "host_common.hpp"

namespace cfft::host
{
// All launchers: asynchronous on `stream`, return hipSuccess or the launch error.
// n = complex frame length handled by the LDS-resident core (16..4096).

// batched complex FFT, interleaved in/out, in may equal out
hipError_t launch_c2c (int n, bool inverse, const float2* in, float2* out, const float2* stage_tw, size_t batch, hipStream_t stream);
} // namespace cfft::host
