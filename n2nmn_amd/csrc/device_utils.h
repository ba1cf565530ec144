// Small device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel, and launchers are
// called from several host threads (PassPipeline workers): set it once per (kernel, device) with an
// atomic bit per device ordinal instead of a process-wide `static bool`.
#include <atomic>
inline void ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

namespace n2nmn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave (64-lane) reductions on the DPP path: quad swaps, half-row / row mirrors, then the two
// row broadcasts; the total lands in lane 63 and is broadcast with v_readlane.  Six VALU
// instructions instead of six dependent ds_bpermute round trips through the LDS hardware
// (__shfl_xor), which made a per-row reduction cost more than streaming the row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL,
                                                    ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float readlane63(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xF>(0.f, v);    // row_half_mirror
  v += dpp_f<0x140, 0xF>(0.f, v);    // row_mirror: every lane holds its 16-lane row's sum
  v += dpp_f<0x142, 0xA>(0.f, v);    // row_bcast15 -> rows 1, 3
  v += dpp_f<0x143, 0xC>(0.f, v);    // row_bcast31 -> rows 2, 3
  return readlane63(v);
}
// Two wave sums for the price of (almost) one: the upper half of `x` and the lower half of `y` change
// places (v_permlane32_swap), one add folds both 64-lane sums into 32 lanes each -- x in lanes 0-31, y
// in lanes 32-63 -- and five DPP steps finish both at once.  Summation order differs from wave_sum's.
__device__ __forceinline__ void wave_sum2(float x, float y, float& sx, float& sy) {
  // (the builtin, not inline asm: hipcc then pads the VALU-write -> permlane-read hazard itself)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  // r[0] = (x[0:31], y[0:31]), r[1] = (x[32:63], y[32:63])
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += dpp_f<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xF>(0.f, v);    // row_half_mirror
  v += dpp_f<0x140, 0xF>(0.f, v);    // row_mirror: every lane holds its 16-lane row's sum
  v += dpp_f<0x142, 0xA>(0.f, v);    // row_bcast15 -> rows 1, 3: lanes 16-31 = sum x, lanes 48-63 = sum y
  sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_f<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f<0x143, 0xC>(v, v));
  return readlane63(v);
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_f<0xB1, 0xF>(v, v));
  v = fminf(v, dpp_f<0x4E, 0xF>(v, v));
  v = fminf(v, dpp_f<0x141, 0xF>(v, v));
  v = fminf(v, dpp_f<0x140, 0xF>(v, v));
  v = fminf(v, dpp_f<0x142, 0xA>(v, v));
  v = fminf(v, dpp_f<0x143, 0xC>(v, v));
  return readlane63(v);
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

// tanh / sigmoid on the hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32 (1 ulp).
//   tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1),  sigmoid(x) = 1 / (1 + 2^(-x log2 e))
// Five / four VALU instructions, two of them quarter-rate.  (`__fdividef` / `__expf` compile to a full
// IEEE division -- v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup -- and an extra
// multiply: 16 instructions per tanh, which made the decoder attention VALU-bound at twice the
// cost.)  Absolute error <= ~3e-7 over the whole range, saturating cleanly to +-1 / 0 / 1 (2^x
// overflows to +inf, whose reciprocal is 0): far inside the 1e-4 logit budget.
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

}  // namespace n2nmn
