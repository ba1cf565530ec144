// Small device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace n2nmn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}

// Block-wide reductions through a caller-provided LDS scratch of >= 16 floats.
// All threads of the block must call; every thread receives the result.
template <int OP>  // 0 sum, 1 max, 2 min
__device__ __forceinline__ float block_reduce(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (OP == 0) v = wave_sum(v); else if (OP == 1) v = wave_max(v); else v = wave_min(v);
  __syncthreads();                 // scratch may still be read from a previous call
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nw; ++i) {
    const float x = scratch[i];
    if (OP == 0) r += x; else if (OP == 1) r = fmaxf(r, x); else r = fminf(r, x);
  }
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace n2nmn
