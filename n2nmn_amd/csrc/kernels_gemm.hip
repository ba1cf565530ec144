// fp32 MFMA GEMM with bias for the dense contractions of the path, plus the operand packers.
//
//   C[M,N] = A[M,K] . B[K,N] + bias           (exact fp32: v_mfma_f32_32x32x2_f32)
//
// Used for: the input-projection tables of the LSTMs (emb . W_x + b), encoder_h_transform
// (models_clevr/nmn3_netgen_att.py:102-106), the hoisted conv_image 1x1 convolution of
// Find / FindSameProperty (util/empty_safe_conv.py:17,29-30 <- nmn3_modules.py:98-99,158-159)
// and the decoder's W_a projection (nmn3_netgen_att.py:185).
//
// CDNA4 mapping
//   * 64x64 block tile, 4 waves (2x2), each wave one 32x32 accumulator (16 VGPR).
//   * Both operands live in LDS in a k-interleaved layout [k/4][row][4]: one ds_read_b128 feeds
//     FOUR consecutive MFMAs.  The MFMA k index is a summation index, so the two half-waves may
//     take any disjoint k's as long as A and B agree: lanes 0-31 take k = 8q..8q+3, lanes 32-63
//     take k = 8q+4..8q+7.
//   * B (weights) is pre-packed in HBM as [Kp/4][Np][4], so its global->LDS copy is a straight,
//     fully coalesced float4 stream; A is read with float4 loads along K (its contiguous axis).
//   * k4-stride in LDS is padded by one float4 so the 8-lane groups of ds_write_b128 (same row,
//     8 different k4) land on 8 distinct bank quads.
//   * register-staged double buffering: global loads of tile t+1 are in flight during the MFMAs
//     of tile t; one barrier per k-tile.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {
constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDS_STRIDE = BM + 1;   // in float4 units

__device__ __forceinline__ void gemm_pk_body(const GemmArgs& a, const int bz, const int bx, const int by) {
  __shared__ float4 As[2][BK / 4][LDS_STRIDE];
  __shared__ float4 Bs[2][BK / 4][LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int M = a.m_dev ? min(a.M, *a.m_dev) : a.M;      // (rows that exist: GemmArgs::m_dev)
  if (m0 >= M) return;
  if (a.gate_tokens) {        // uniform early exit: none of this tile's images needs the map
    __shared__ int need;
    if (tid == 0) need = 0;
    __syncthreads();
    const int g0 = m0 / a.gate_rows, g1 = min(m0 + BM - 1, M - 1) / a.gate_rows;
    const int per = a.gate_T;
    if (tid < (g1 - g0 + 1) * per) {
      const int g = g0 + tid / per, t = tid % per;
      const int tok = a.gate_tokens[(size_t)t * a.gate_N + g];
      if (tok >= 0 && tok < a.gate_V && a.gate_token_op[tok] == a.gate_op) need = 1;
    }
    __syncthreads();
    if (!need) return;
  }

  // ---- global load assignment: 2 float4 of A and 2 of B per thread -----------------------
  // A: float4 index i = tid + 256 j -> row = i >> 3, k4 = i & 7 (8 lanes cover 128 contiguous B)
  // B: float4 index i -> k4 = i >> 6, n = i & 63 (one contiguous KiB per k4)
  const int a_row0 = tid >> 3, a_row1 = (tid + 256) >> 3;
  const int a_k40 = tid & 7, a_k41 = a_k40;
  auto src_row = [&](int r) {
    int src = r < M ? r : 0;
    if (a.group_idx) {
      const int g = src / a.group_size;
      src = a.group_idx[g] * a.group_size + (src - g * a.group_size);
    }
    return src;
  };
  const bool a_ok0 = m0 + a_row0 < M, a_ok1 = m0 + a_row1 < M;
  const float* a_ptr0 = a.A + (size_t)src_row(m0 + a_row0) * a.lda;
  const float* a_ptr1 = a.A + (size_t)src_row(m0 + a_row1) * a.lda;
  const int b_k40 = tid >> 6, b_k41 = (tid + 256) >> 6, b_n0 = tid & 63, b_n1 = b_n0;
  const float4* b_ptr0 = reinterpret_cast<const float4*>(a.Bp) + (size_t)b_k40 * a.Np + n0 + b_n0;
  const float4* b_ptr1 = reinterpret_cast<const float4*>(a.Bp) + (size_t)b_k41 * a.Np + n0 + b_n1;

  // split-K (ksplit > 1, accumulate only): bz owns a contiguous range of k-tiles and adds
  // its partial tile atomically -- for the skinny NT products of the backward pass ([V,4L].[4L,E])
  int ktb = 0, nkt = a.Kp / BK;
  if (a.ksplit > 1) {
    const int per = (nkt + a.ksplit - 1) / a.ksplit;
    ktb = bz * per;
    nkt = min(per, nkt - ktb);
    if (nkt <= 0) return;
  }
  const size_t bstep = (size_t)(BK / 4) * a.Np;     // float4 stride of one k-tile in packed B
  // register-staged pipeline, two k-tiles deep: tile kt+2 is being fetched while tile kt+1 sits
  // in registers and tile kt is consumed from LDS (a global round trip spans two MFMA phases)
  float4 ra0, ra1, rb0, rb1;     // staging set X
  float4 sa0, sa1, sb0, sb1;     // staging set Y
// loads are unconditional from a clamped (always valid) address and masked afterwards: a
// conditional load would be lowered to a flat load through a select with a scratch zero
#define N2_GLOAD(KT, A0, A1, B0, B1)                                                        \
  do {                                                                                      \
    const int kk0 = (ktb + (KT)) * BK + 4 * a_k40, kk1 = (ktb + (KT)) * BK + 4 * a_k41;     \
    const float4 t0 = *reinterpret_cast<const float4*>(a_ptr0 + (kk0 < a.K ? kk0 : a.K - 4)); \
    const float4 t1 = *reinterpret_cast<const float4*>(a_ptr1 + (kk1 < a.K ? kk1 : a.K - 4)); \
    const bool k0 = a_ok0 && kk0 < a.K, k1 = a_ok1 && kk1 < a.K;                            \
    A0.x = k0 ? t0.x : 0.f; A0.y = k0 ? t0.y : 0.f; A0.z = k0 ? t0.z : 0.f; A0.w = k0 ? t0.w : 0.f; \
    A1.x = k1 ? t1.x : 0.f; A1.y = k1 ? t1.y : 0.f; A1.z = k1 ? t1.z : 0.f; A1.w = k1 ? t1.w : 0.f; \
    B0 = b_ptr0[(size_t)(ktb + (KT)) * bstep];                                              \
    B1 = b_ptr1[(size_t)(ktb + (KT)) * bstep];                                              \
  } while (0)
#define N2_LSTORE(BUF, A0, A1, B0, B1)                                                      \
  do {                                                                                      \
    As[BUF][a_k40][a_row0] = A0; As[BUF][a_k41][a_row1] = A1;                       \
    Bs[BUF][b_k40][b_n0] = B0; Bs[BUF][b_k41][b_n1] = B1;                           \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;

  N2_GLOAD(0, ra0, ra1, rb0, rb1);
  if (nkt > 1) N2_GLOAD(1, sa0, sa1, sb0, sb1);
  N2_LSTORE(0, ra0, ra1, rb0, rb1);
  __syncthreads();
  // invariant at the top of iteration kt (cur = kt & 1): LDS[cur] holds tile kt; set Y holds
  // tile kt+1 (if any); set X is free
  for (int kt = 0; kt < nkt; kt += 2) {
    if (kt + 2 < nkt) N2_GLOAD(kt + 2, ra0, ra1, rb0, rb1);
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      const float4 av = As[0][2 * kq + kh][wm * 32 + li];
      const float4 bv = Bs[0][2 * kq + kh][wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    if (kt + 1 < nkt) N2_LSTORE(1, sa0, sa1, sb0, sb1);
    __syncthreads();
    if (kt + 1 >= nkt) break;
    if (kt + 3 < nkt) N2_GLOAD(kt + 3, sa0, sa1, sb0, sb1);
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      const float4 av = As[1][2 * kq + kh][wm * 32 + li];
      const float4 bv = Bs[1][2 * kq + kh][wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    if (kt + 2 < nkt) N2_LSTORE(0, ra0, ra1, rb0, rb1);
    __syncthreads();
  }
#undef N2_GLOAD
#undef N2_LSTORE

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col = n0 + wn * 32 + li;
  if (col < a.n_store) {
    const float bias = (a.bias && col < a.N && bz == 0) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M) {
        float val = (col < a.N) ? acc[r] + bias : 0.f;
        if (a.relu) val = fmaxf(val, 0.f);
        int orow = row;
        if (a.c_row_idx) {
          orow = a.c_row_idx[row];
          if (orow < 0) continue;
        }
        float* dst = a.C + (size_t)orow * a.ldc + col;
        if (a.ksplit > 1) atomicAdd(dst, val);
        else *dst = a.accumulate ? *dst + val : val;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_pk_kernel(GemmArgs a) {
  gemm_pk_body(a, blockIdx.z, blockIdx.x, blockIdx.y);
}

// two independent problems of the same tile shape in one launch (blockIdx.z selects): the hoisted
// conv_image GEMMs of FindModule and (token-gated) FindSamePropertyModule share one grid, so the
// gated problem's few live tiles fill the first problem's tail instead of paying a launch of their own
__global__ __launch_bounds__(256) void gemm_pk2_kernel(GemmArgs a0, GemmArgs a1) {
  const GemmArgs a = blockIdx.z == 0 ? a0 : a1;
  if ((int)blockIdx.y * BM >= a.M || (int)blockIdx.x * BN >= a.Np) return;
  gemm_pk_body(a, 0, blockIdx.x, blockIdx.y);
}

// Up to four independent problems of this tile shape in ONE launch over a flat tile list (no
// split-K).  Small GEMMs of the forward pass (encoder_h_transform 360 tiles, q 160, conv_image
// 600 + the gated FindSameProperty tiles) each fill the 256 CUs for one and a fraction rounds; as one
// list the fractions add up instead of each paying a partly empty last round and a launch.
// XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2),
// so id -> (xcd = id % 8, slot = id / 8) is mapped to list position ((slot / 8) * 8 + xcd) * 8 +
// slot % 8: runs of 8 consecutive positions -- the column tiles of one or two row tiles, which
// share their A rows -- execute on ONE XCD and fetch those rows into one L2 instead of up to
// eight, while the runs themselves rotate over the XCDs so that every XCD sees the same mix of
// problems (a contiguous eighth of the list per XCD left the XCDs holding the gated problem idle).
__global__ __launch_bounds__(256) void gemm_pkn_kernel(GemmBatch b) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = (((slot >> 3) << 3) + xcd) * 8 + (slot & 7);
  if (t >= b.start[4]) return;
  const int p = (t >= b.start[1]) + (t >= b.start[2]) + (t >= b.start[3]);
  const GemmArgs a = p == 0 ? b.a[0] : p == 1 ? b.a[1] : p == 2 ? b.a[2] : b.a[3];
  const int local = t - (p == 0 ? 0 : p == 1 ? b.start[1] : p == 2 ? b.start[2] : b.start[3]);
  const int gx = (a.n_store + BN - 1) / BN;
  gemm_pk_body(a, 0, local % gx, local / gx);
}

__global__ void pack_pk_kernel(const float* __restrict__ src, int ld, int K, int N,
                               float* __restrict__ dst, int Kp, int Np) {
  const size_t total = (size_t)Kp * Np;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 3);
    const size_t r = i >> 2;
    const int n = (int)(r % Np);
    const int k = (int)(r / Np) * 4 + kk;
    dst[i] = (k < K && n < N) ? src[(size_t)k * ld + n] : 0.f;
  }
}

// dst[tile j][k/4][c][k%4] = W[row0 + k][col(j, c)],  c = 0..15
//   gate_L > 0: col = (c>>2)*gate_L + 4j + (c&3)   (i,j,f,o gates of hidden units 4j..4j+3)
//   gate_L == 0: col = 16j + c
__global__ void pack_tiles_kernel(const float* __restrict__ W, int ld, int row0, int K, int ntiles,
                                  int gate_L, float* __restrict__ dst) {
  const size_t total = (size_t)ntiles * K * 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 3);
    const int c = (int)((i >> 2) & 15);
    const size_t r = i >> 6;
    const int k4 = (int)(r % (K / 4));
    const int j = (int)(r / (K / 4));
    const int k = 4 * k4 + kk;
    const int col = gate_L > 0 ? (c >> 2) * gate_L + 4 * j + (c & 3) : 16 * j + c;
    dst[i] = W[(size_t)(row0 + k) * ld + col];
  }
}
__global__ void pad_rows_kernel(const float* __restrict__ src, int R, int M,
                                float* __restrict__ dst, int Mp) {
  const size_t total = (size_t)R * Mp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Mp);
    const size_t r = i / Mp;
    dst[i] = c < M ? src[r * M + c] : 0.f;
  }
}
// one workgroup = PACK_ELEMS_PER_BLOCK consecutive elements of one job (same index maps as the
// stand-alone packers above and pack_pk_t / pack_tiles_t in kernels_train.hip)
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs,
                                                        int njobs) {
  __shared__ int sj;
  if (threadIdx.x == 0) {
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].block0 <= blockIdx.x) ++j;
    sj = j;
  }
  __syncthreads();
  const PackJob jb = jobs[sj];
  const uint32_t base = (uint32_t)(blockIdx.x - jb.block0) * PACK_ELEMS_PER_BLOCK;
  // 32-bit index arithmetic (every job has < 2^31 elements; the 64-bit divisions of the first
  // version were most of the kernel), and the 16 elements of a thread are gathered in two groups
  // of 8 independent loads before anything is stored
  constexpr int PER = PACK_ELEMS_PER_BLOCK / 256, U = 8;
  static_assert(PER % U == 0, "whole groups");
  // element i -> (source offset or -1 for a zero, destination offset)
  auto map = [&](uint32_t i, int64_t& so, uint32_t& dofs) {
    dofs = i;
    switch (jb.kind) {
      case PJ_PK:
      case PJ_PK_T: {
        const uint32_t ld = jb.p[0], K = jb.p[1], N = jb.p[2], Np = jb.p[4];
        const uint32_t kk = i & 3, r = i >> 2;
        const uint32_t q = r / Np, n = r - q * Np;
        const uint32_t kq = q * 4 + kk;
        so = (kq < K && n < N) ? (jb.kind == PJ_PK ? (int64_t)kq * ld + n : (int64_t)n * ld + kq)
                               : -1;
        break;
      }
      case PJ_PK_GATES: {     // PK pack whose output columns are in tile order (see LstmJob::xtab):
        const uint32_t ld = jb.p[0], K = jb.p[1], L = jb.p[2], Np = 4 * L;   // n' = 16 tile + 4 gate + unit
        const uint32_t kk = i & 3, r = i >> 2;
        const uint32_t q = r / Np, n = r - q * Np;
        const uint32_t kq = q * 4 + kk;
        const uint32_t col = ((n >> 2) & 3) * L + 4 * (n >> 4) + (n & 3);
        so = kq < K ? (int64_t)kq * ld + col : -1;
        break;
      }
      case PJ_VEC_GATES: {    // a [4L] vector (bias) in the same column order
        const uint32_t L = jb.p[0];
        so = (int64_t)(((i >> 2) & 3) * L + 4 * (i >> 4) + (i & 3));
        break;
      }
      case PJ_TILES: {
        const uint32_t ld = jb.p[0], row0 = jb.p[1], K = jb.p[2], gate_L = jb.p[3];
        const uint32_t kk = i & 3, c = (i >> 2) & 15, r = i >> 6;
        const uint32_t K4 = K >> 2;
        const uint32_t j = r / K4, k4 = r - j * K4;
        const uint32_t col = gate_L > 0 ? (c >> 2) * gate_L + 4 * j + (c & 3) : 16 * j + c;
        so = (int64_t)(row0 + 4 * k4 + kk) * ld + col;
        break;
      }
      case PJ_TILES64: {      // 16-unit tiles of lstm_tile_kernel: [L/16][K/4][64][4], c = 16 * gate + unit
        const uint32_t ld = jb.p[0], row0 = jb.p[1], K = jb.p[2], L = jb.p[3];
        const uint32_t kk = i & 3, c = (i >> 2) & 63, r = i >> 8;
        const uint32_t K4 = K >> 2;
        const uint32_t j = r / K4, k4 = r - j * K4;
        so = (int64_t)(row0 + 4 * k4 + kk) * ld + (c >> 4) * L + 16 * j + (c & 15);
        break;
      }
      case PJ_TILES_T: {
        const uint32_t ld = jb.p[0], row0 = jb.p[1], L = jb.p[2], Ktot = jb.p[3], k_off = jb.p[4];
        const uint32_t gi = i & 3, c = (i >> 2) & 15, r = i >> 6;
        const uint32_t j = r / L, u = r - j * L;
        dofs = ((j * (Ktot >> 2) + (k_off >> 2) + u) * 16 + c) * 4 + gi;
        so = (int64_t)(row0 + 16 * j + c) * ld + (int64_t)gi * L + u;
        break;
      }
      default: {   // PJ_PAD
        const uint32_t M = jb.p[1], Mp = jb.p[2];
        const uint32_t r = i / Mp, c = i - r * Mp;
        so = c < M ? (int64_t)r * M + c : -1;
        break;
      }
    }
  };
  for (int g0 = 0; g0 < PER; g0 += U) {
    float v[U];
    uint32_t dofs[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + threadIdx.x + 256 * (g0 + u);
      on[u] = i < jb.total;
      int64_t so = -1;
      dofs[u] = 0;
      if (on[u]) map(i, so, dofs[u]);
      v[u] = jb.src[so >= 0 ? so : 0];          // unconditional load from a valid address
      if (so < 0) v[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (on[u]) jb.dst[dofs[u]] = v[u];
  }
}

}  // namespace

void launch_pack_jobs(const PackJob* jobs_dev, int njobs, int total_blocks, hipStream_t s) {
  if (njobs <= 0 || total_blocks <= 0) return;
  hipLaunchKernelGGL(pack_jobs_kernel, dim3(total_blocks), dim3(256), 0, s, jobs_dev, njobs);
}

void launch_pad_rows(const float* src, int R, int M, float* dst, int Mp, hipStream_t s) {
  const size_t total = (size_t)R * Mp;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, s, src, R, M, dst, Mp);
}

void launch_gemm_pk2(const GemmArgs& a0, const GemmArgs& a1, hipStream_t s) {
  const int gx = std::max((a0.n_store + BN - 1) / BN, (a1.n_store + BN - 1) / BN);
  const int gy = std::max((a0.M + BM - 1) / BM, (a1.M + BM - 1) / BM);
  hipLaunchKernelGGL(gemm_pk2_kernel, dim3(gx, gy, 2), dim3(256), 0, s, a0, a1);
}

// The 128 x 128 LDS-DMA tiles win once a launch has enough of them to cover the chip; below that the
// 64 x 64 tiles of gemm_pk / gemm_pkn spread the same work over four times as many workgroups (a
// training step's encoder_h_transform, 2880 x 512 x 512 = 92 big tiles: 39.6 us against 27.8).
static bool use_gemm_dma(const GemmArgs* a, int n) {
  static const int on = N2NMN_KNOB_INT("N2NMN_GEMM_DMA", 1);
  static const int min_tiles = N2NMN_KNOB_INT("N2NMN_GEMM_DMA_MIN_TILES", 192);
  if (!on) return false;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (a[i].M <= 0) continue;
    if (!gemm_dma_supported(a[i])) return false;
    tiles += ((a[i].M + 127) / 128) * ((a[i].n_store + 127) / 128);
  }
  return tiles >= min_tiles;
}

// gemm_dma3_kernel (the dense contractions of the opt-in bf16x3 mode on split operands, rounds 4 - 5) is NOT part of
// this library.  With a second stream running passes concurrently, passes whose conv_image launch ran on it returned
// wrong logits (1e-5 .. 1e-2, 30 - 50 % of the rounds).  Round 6 (profiles/r06_notes.md section 1) cleared the kernel's
// own data path -- every LDS-DMA piece verified in LDS, inputs and the output buffer checksummed right -- and found the
// fault between the launch and its consumer: a kernel launched directly behind it computes from wrong values, a ~40 us
// idle gap behind it removes the fault, a release fence or a drained store queue at the end of the kernel does not.
// Unexplained at that level, so the kernel left the product; it lives in tools/diag/csrc/ and is compiled only into
// the diagnostic library (tools/diag/build_diag.py, -DN2NMN_DIAG), where N2NMN_GEMM_DMA3=1 routes launches to it.
#ifdef N2NMN_DIAG
static bool use_gemm_dma3(const GemmArgs* a, int n) {
  static const int on = N2NMN_KNOB_INT("N2NMN_GEMM_DMA3", 0);
  if (!on) return false;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (a[i].M <= 0) continue;
    if (!gemm_dma3_supported(a[i])) return false;
    tiles += ((a[i].M + 63) / 64) * ((a[i].n_store + 127) / 128);
  }
  return tiles >= 256;
}
#else
static bool use_gemm_dma3(const GemmArgs*, int) { return false; }
bool gemm_dma3_supported(const GemmArgs&) { return false; }
void launch_gemm_dma3(const GemmArgs*, int, hipStream_t) {}
void launch_pack_pk_b3(const float*, int, int, uint16_t*, hipStream_t) {}
#endif

void launch_gemm_pkn(const GemmArgs* a, int n, hipStream_t s) {
  if (use_gemm_dma3(a, n)) { launch_gemm_dma3(a, n, s); return; }
  if (use_gemm_dma(a, n)) { launch_gemm_dma(a, n, s); return; }
  GemmBatch b{};
  int tiles = 0, np = 0;
  for (int i = 0; i < n && np < 4; ++i) {
    if (a[i].M <= 0) continue;
    b.a[np] = a[i];
    b.start[np] = tiles;
    tiles += ((a[i].n_store + BN - 1) / BN) * ((a[i].M + BM - 1) / BM);
    ++np;
  }
  if (!np) return;
  for (int i = np; i <= 4; ++i) b.start[i] = tiles;
  hipLaunchKernelGGL(gemm_pkn_kernel, dim3((tiles + 63) / 64 * 64), dim3(256), 0, s, b);
}

void launch_gemm_pk(const GemmArgs& a, hipStream_t s) {
  if (use_gemm_dma3(&a, 1)) { launch_gemm_dma3(&a, 1, s); return; }
  if (use_gemm_dma(&a, 1)) { launch_gemm_dma(&a, 1, s); return; }
  dim3 grid((a.n_store + BN - 1) / BN, (a.M + BM - 1) / BM, a.ksplit > 1 ? a.ksplit : 1);
  if (a.M <= 0) return;
  hipLaunchKernelGGL(gemm_pk_kernel, grid, dim3(256), 0, s, a);
}

void launch_pack_pk(const float* src, int ld, int K, int N, float* dst, int Kp, int Np,
                    hipStream_t s) {
  const size_t total = (size_t)Kp * Np;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_pk_kernel, dim3(blocks), dim3(256), 0, s, src, ld, K, N, dst, Kp, Np);
}

void launch_pack_tiles(const float* W, int ld, int row0, int K, int ntiles, int gate_L, float* dst,
                       hipStream_t s) {
  const size_t total = (size_t)ntiles * K * 16;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_tiles_kernel, dim3(blocks), dim3(256), 0, s, W, ld, row0, K, ntiles,
                     gate_L, dst);
}

}  // namespace n2nmn
