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

// A/B switches of the launchers.  The product library reads NO environment variable: each switch is its
// default, fixed at compile time.  Only the diagnostic build (tools/diag/build_diag.py, -DN2NMN_DIAG) looks
// the name up in the environment (once, on first use).  Per-context switches that tests need at run time go
// through n2nmn_debug_set (include/n2nmn.h section 7).
#ifdef N2NMN_DIAG
#include <cstdlib>
#define N2NMN_KNOB_INT(name, dflt) ([] { const char* e_ = std::getenv(name); return e_ ? std::atoi(e_) : (dflt); }())
#else
#define N2NMN_KNOB_INT(name, dflt) (dflt)
#endif

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


// ---- split-operand bf16 (kernels_lstm_tile3.hip): an fp32 value as the exact sum of three bf16 ----------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
// x = hi + mid + lo, each a bf16 (round to nearest even; the two subtractions are exact)
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const f32x2v v = {x0, x1};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  const f32x2v r1 = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  mid = __builtin_bit_cast(uint32_t, m);
  const f32x2v r2 = {r1[0] - __uint_as_float(mid << 16), r1[1] - __uint_as_float(mid & 0xffff0000u)};
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  lo = __builtin_bit_cast(uint32_t, l);
}

// the three planes of four consecutive hidden units (a float4 of the state) -> 8 bytes per plane
__device__ __forceinline__ void store_planes(uint16_t* planes, size_t plane_elems, size_t off, float4 h) {
  uint32_t a[3], b[3];
  split3(h.x, h.y, a[0], a[1], a[2]);
  split3(h.z, h.w, b[0], b[1], b[2]);
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<uint2*>(planes + (size_t)p * plane_elems + off) = make_uint2(a[p], b[p]);
}


// one element (hidden unit `unit` of state row `row`) of the planes [3][L/8][R][8] of a state buffer
__device__ __forceinline__ void store_plane1(uint16_t* planes, int L, int R, int unit, int row, float x) {
  uint32_t hi, mid, lo;
  split3(x, 0.f, hi, mid, lo);
  const size_t plane_elems = (size_t)(L / 8) * R * 8;
  const size_t off = ((size_t)(unit >> 3) * R + row) * 8 + (unit & 7);
  planes[off] = (uint16_t)hi; planes[plane_elems + off] = (uint16_t)mid; planes[2 * plane_elems + off] = (uint16_t)lo;
}

}  // namespace n2nmn
