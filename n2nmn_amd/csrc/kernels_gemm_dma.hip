// gemm_dma_kernel: C[M,N] = A[M,K] . B[K,N] + bias (exact fp32, v_mfma_f32_32x32x2_f32) for the large
// dense contractions of a pass -- encoder_h_transform (models_clevr/nmn3_netgen_att.py:102-106), the
// decoder's W_a projection (:185) and the hoisted conv_image 1x1 convolutions of Find /
// FindSameProperty (util/empty_safe_conv.py:17,29-30 <- nmn3_modules.py:98-99,158-159).
//
// gemm_pk_kernel (64 x 64 tile, operands staged through registers, one barrier per 32 k) runs these
// at 92-94 TFLOP/s; the merged launch is 19 % of a pass.  This kernel applies what the recurrent step
// (kernels_lstm_tile.hip) showed to work on this chip:
//   * 128 x 128 workgroup tile, 8 waves (4 x 2), each wave 32 rows x 64 columns = two 32x32
//     accumulators that share their A fragment: half the L2 -> CU operand bytes per flop of the 64 x 64
//     tile and 3 LDS reads per 8 MFMAs;
//   * both operands reach LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no
//     VGPRs, no ds_write pass), two 32 KiB stages of 32 k, the DMA of stage i + 1 in flight under the
//     whole of stage i; counted s_waitcnt vmcnt + ONE raw s_barrier per stage;
//   * B (weights) is the PK pack [Kp/4][Np][4]: a k4 slab of the tile is 2 KiB contiguous and its LDS
//     image is read conflict-free as it lies.  A is row-major in HBM (features [rows][D], encoder
//     outputs [T N][L]): a DMA instruction copies 128 contiguous bytes (32 k) of 8 rows, so the LDS
//     image is [row][8 chunks of 16 B], and a 32-row x 1-chunk MFMA fragment read would hit two bank
//     quads 16 times each.  LDS-DMA writes lane-linearly, so the swizzle is applied on the SOURCE side
//     (cdna_hip_programming.md rule 21): LDS position c of a row holds chunk c ^ ((row >> 1) & 7), the
//     fragment read applies the same XOR, and the 16 lanes of every ds_read_b128 group then cover 16
//     distinct bank quads;
//   * operand fragments are double-buffered in registers inside a stage; the first fragment of a
//     stage is read after the barrier (two workgroups per CU = 4 waves per SIMD cover it).
// Launch: up to four problems as one flat, XCD-rotated tile list (as launch_gemm_pkn).
#include <algorithm>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

constexpr int DM = 128, DN = 128, DK = 32;
constexpr int DMA_THREADS = 512;
constexpr int DSTAGE = (DM * DK + DK * DN) * 4;          // 32 KiB: A image 16 KiB, then B image
constexpr int DB_IMAGE = DM * DK * 4;

__device__ __forceinline__ void glds16(const float* base, uint32_t voff, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(base), "s"(lds)
      : "memory");
}

struct Frag { float4 a; float4 b[2]; };

__device__ __forceinline__ void gemm_dma_body(const GemmArgs& a, const int bx, const int by) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * DM, n0 = bx * DN;
  const int M = a.m_dev ? min(a.M, *a.m_dev) : a.M;      // (rows that exist: GemmArgs::m_dev)
  if (m0 >= M) return;
  if (a.gate_tokens) {        // uniform early exit: none of this tile's images needs the map
    __shared__ int need;
    if (tid == 0) need = 0;
    __syncthreads();
    const int g0 = m0 / a.gate_rows, g1 = min(m0 + DM - 1, M - 1) / a.gate_rows;
    const int per = a.gate_T;
    for (int i = tid; i < (g1 - g0 + 1) * per; i += DMA_THREADS) {
      const int g = g0 + i / per, t = i % per;
      const int tok = a.gate_tokens[(size_t)t * a.gate_N + g];
      if (tok >= 0 && tok < a.gate_V && a.gate_token_op[tok] == a.gate_op) need = 1;
    }
    __syncthreads();
    if (!need) return;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int nst = a.Kp / DK;

  // ---- DMA roles: wave w moves A rows [16w, 16w + 16) (two pieces of 8 rows x 128 B) and the k4
  // slab w of B (two pieces of 64 columns) of every stage ------------------------------------------
  uint32_t a_off[2], a_sw[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 16 * w + 8 * u + (lane >> 3);
    int gm = m0 + row;
    gm = gm < M ? gm : M - 1;
    if (a.group_idx) {
      const int g = gm / a.group_size;
      gm = a.group_idx[g] * a.group_size + (gm - g * a.group_size);
    }
    a_off[u] = (uint32_t)gm * (uint32_t)a.lda * 4u;
    a_sw[u] = (uint32_t)((lane & 7) ^ ((row >> 1) & 7));      // chunk this lane fetches
  }
  const float* const Ap = a.A;
  const float* const Bp = a.Bp;
  const uint32_t b_off = ((uint32_t)w * (uint32_t)a.Np + (uint32_t)n0 + (uint32_t)lane) * 16u;
  const uint32_t b_slab = 8u * (uint32_t)a.Np * 16u;            // bytes per stage of B (8 k4 slabs)
  const int Klast = a.K - 4;
  auto issue = [&](int s) {
    const uint32_t st = lds0 + (uint32_t)(s & 1) * DSTAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // columns past K (Kp padding) meet zero weights: any finite in-bounds value will do
      const int k = min(s * DK + 4 * (int)a_sw[u], Klast);
      glds16(Ap, a_off[u] + (uint32_t)k * 4u, st + (uint32_t)(16 * w + 8 * u) * 128u);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      glds16(Bp, b_off + (uint32_t)s * b_slab + (uint32_t)u * 1024u,
             st + DB_IMAGE + (uint32_t)(w * DN + 64 * u) * 16u);
  };

  // ---- fragment addressing -------------------------------------------------------------------------
  const int li = lane & 31, kh = lane >> 5;
  const int arow = wm * 32 + li;
  const uint32_t a_base = (uint32_t)arow * 128u;
  const uint32_t a_x = (uint32_t)((arow >> 1) & 7);
  const uint32_t b_base = (uint32_t)DB_IMAGE + (uint32_t)(wn * 64 + li) * 16u;
  auto read_frag = [&](int s, int g) {
    const char* st = smem + (size_t)(s & 1) * DSTAGE;
    const uint32_t chunk = (uint32_t)(2 * g + kh);
    Frag f;
    f.a = *reinterpret_cast<const float4*>(st + a_base + ((chunk ^ a_x) << 4));
    f.b[0] = *reinterpret_cast<const float4*>(st + b_base + chunk * (DN * 16u));
    f.b[1] = *reinterpret_cast<const float4*>(st + b_base + chunk * (DN * 16u) + 32u * 16u);
    return f;
  };
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.x, f.b[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.y, f.b[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.z, f.b[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.w, f.b[t].w, acc[t], 0, 0, 0);
  };

  issue(0);
  for (int s = 0; s < nst; ++s) {
    // this wave's pieces of stage s have landed; after the barrier everyone's have, and everyone is
    // done reading stage s - 1, whose buffer the DMA of stage s + 1 refills under this stage's MFMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < nst) issue(s + 1);
    Frag P = read_frag(s, 0);
    Frag Q = read_frag(s, 1);
    __builtin_amdgcn_sched_barrier(0);       // (left alone, hipcc sinks the reads below the MFMAs)
    mma(P);
    P = read_frag(s, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(Q);
    Q = read_frag(s, 3);
    __builtin_amdgcn_sched_barrier(0);
    mma(P);
    mma(Q);
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = n0 + wn * 64 + 32 * t + li;
    if (col >= a.n_store) continue;
    const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M) {
        float val = (col < a.N) ? acc[t][r] + bias : 0.f;
        if (a.relu) val = fmaxf(val, 0.f);
        int orow = row;
        if (a.c_row_idx) {
          orow = a.c_row_idx[row];
          if (orow < 0) continue;
        }
        float* dst = a.C + (size_t)orow * a.ldc + col;
        *dst = a.accumulate ? *dst + val : val;
      }
    }
  }
}

__global__ __launch_bounds__(DMA_THREADS) void gemm_dma_kernel(GemmBatch b) {
  // consecutive workgroup ids go round-robin over the 8 XCDs: runs of 8 consecutive list positions
  // (the column tiles of 2-4 row tiles, which share their A rows) execute on ONE XCD, the runs rotate
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = (((slot >> 3) << 3) + xcd) * 8 + (slot & 7);
  if (t >= b.start[4]) return;
  const int p = (t >= b.start[1]) + (t >= b.start[2]) + (t >= b.start[3]);
  const GemmArgs& a = b.a[p];
  const int local = t - b.start[p];
  const int gx = (a.n_store + DN - 1) / DN;
  gemm_dma_body(a, local % gx, local / gx);
}

}  // namespace

bool gemm_dma_supported(const GemmArgs& a) {
  // A is addressed with 32-bit byte offsets from its base: the bound is on the rows that can be
  // READ -- with a row gather (group_idx) those of the source table, which M does not bound
  const size_t a_rows = a.group_idx ? (size_t)a.src_rows : (size_t)a.M;
  if (a.group_idx && a.src_rows <= 0) return false;
  return a.M >= DM && a.Np % DN == 0 && a.Kp % DK == 0 && a.K % 4 == 0 && a.K >= 4 && a.lda % 4 == 0 &&
         a.ksplit <= 1 && a_rows * a.lda * 4 < ((size_t)1 << 32) &&
         (size_t)a.Kp * a.Np * 4 < ((size_t)1 << 32) && (!a.gate_tokens || a.gate_T <= 64);
}

void launch_gemm_dma(const GemmArgs* a, int n, hipStream_t s) {
  GemmBatch b{};
  int tiles = 0, np = 0;
  for (int i = 0; i < n && np < 4; ++i) {
    if (a[i].M <= 0) continue;
    b.a[np] = a[i];
    b.start[np] = tiles;
    tiles += ((a[i].n_store + DN - 1) / DN) * ((a[i].M + DM - 1) / DM);
    ++np;
  }
  if (!np) return;
  for (int i = np; i <= 4; ++i) b.start[i] = tiles;
  static std::atomic<uint64_t> attr{0};
  ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_dma_kernel), 2 * DSTAGE, attr);
  hipLaunchKernelGGL(gemm_dma_kernel, dim3((tiles + 63) / 64 * 64), dim3(DMA_THREADS), 2 * DSTAGE, s, b);
}

}  // namespace n2nmn
