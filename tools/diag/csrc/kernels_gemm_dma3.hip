// gemm_dma3_kernel: the dense contractions of a pass (encoder_h_transform, the decoder's W_a projection,
// the hoisted conv_image 1x1 convolutions: see kernels_gemm_dma.hip) in the OPT-IN split-operand bf16 mode
// (N2NMN_MODE_THROUGHPUT_BF16X3, kernels_lstm_tile3.hip): x = hi + mid + lo exactly (three bf16, round to
// nearest, exact residuals), 6 of the 9 cross products on v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//
//   * B (weights) is split ONCE per commit (pack_pk_b3_kernel) into fragment-ordered planes
//     [K/32][Np/128][3 planes][8 column tiles][64 lanes][8 bf16]: a stage of a 128-column tile is 24
//     contiguous KiB, moved by LDS-DMA and read back as it lies (ds_read_b128 per plane and column tile);
//   * A (activations: image features, encoder outputs, decoder outputs) stays fp32 in HBM and reaches LDS
//     by LDS-DMA exactly as in gemm_dma_kernel (row-major [row][8 chunks of 16 B], swizzled on the source
//     side); a wave reads the 8 consecutive k of its row (two chunks), splits them in registers (5.5 VALU
//     instructions per element, once per 24 MFMAs) and feeds the matrix cores directly -- a pre-split A
//     would cost an extra pass over the activations;
//   * workgroup tile 64 rows x 128 columns, 8 waves (4 row tiles x 2 column halves), two stages of 32 KiB:
//     two workgroups per CU.  A wave: 1 A fragment (3 planes), 4 column tiles x 3 planes of B, 24 MFMAs
//     per 32 k.
// Gather / scatter / device row count / token gate as in gemm_dma_kernel.
#include <algorithm>
#ifdef N2NMN_DIAG_SUMS
#include <mutex>
#include <vector>
#endif

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MT = 16-row tiles per wave: workgroup tile 64 MT rows x 128 columns (MT = 2: every B fragment read from
// LDS serves two row tiles -- the 64-row tile spends as long reading LDS as on the matrix cores)
constexpr int N3 = 128, K3 = 32;
constexpr int G3_THREADS = 512;
constexpr int B3_IMAGE = 3 * N3 * K3 * 2;                // 24 KiB: [plane][column tile][lane][16 B]
template <int MT> struct G3 {
  static constexpr int M = 64 * MT;
  static constexpr int A_IMAGE = M * K3 * 4;             // 8 KiB fp32 per 64 rows
  static constexpr int STAGE = A_IMAGE + B3_IMAGE;       // 32 / 40 KiB
};

__device__ __forceinline__ void glds16_3(const void* base, uint32_t voff, uint32_t lds) {
  uint32_t keep;
#ifdef N2NMN_DIAG_TWICE
  lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);   // (the pass loop leaves the address in a VGPR)
#endif
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


#ifdef N2NMN_DIAG
// ---- diagnostic build only (tools/diag/build_diag.py, -DN2NMN_DIAG): in-kernel verification of what the
// LDS-DMA left in LDS against a direct global read of the same bytes, with the workgroup's LDS allocation and
// hardware ids recorded per anomaly (profiles/r06_notes.md section 1).  Nothing of this is in the product .so.
struct DiagRec { uint32_t w[16]; };
__device__ DiagRec* g_diag_rec = nullptr;
__device__ uint32_t g_diag_cap = 0;
__device__ uint32_t* g_diag_cnt = nullptr;        // [0] records wanted, [1] workgroups seen, [2..] misc
__device__ uint32_t g_diag_level = 0;
__device__ uint32_t* g_diag_wg = nullptr;         // optional: per-workgroup {lds_alloc, hw_id, xcc_id, tile}
__device__ uint32_t g_diag_wg_cap = 0;

__device__ __forceinline__ uint32_t diag_lds_alloc() {
  uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(v)); return v;
}
__device__ __forceinline__ uint32_t diag_hw_id() {
  uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v;
}
__device__ __forceinline__ uint32_t diag_xcc_id() {
  uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v;
}
typedef uint32_t diag_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 diag_lds_read(uint32_t lds_addr) {      // (an LDS read the compiler cannot move)
  diag_u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ bool diag_ne(const uint4& a, const uint4& b) {
  return a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
}
// kind: 1 = own piece wrong right behind this wave's vmcnt(0); 2 = another wave's piece wrong behind the
// barrier; 3 = the two redundant accumulator chains differ
__device__ __forceinline__ void diag_record(uint32_t kind, uint32_t tile, uint32_t stage, uint32_t wave,
                                            uint32_t piece, uint64_t mask, uint32_t got, uint32_t exp,
                                            uint32_t lds_off, uint32_t reread_ok, uint32_t stale_match,
                                            uint32_t extra) {
  if (!g_diag_cnt) return;
  const uint32_t i = atomicAdd(&g_diag_cnt[0], 1u);
  if (i >= g_diag_cap) return;
  DiagRec r;
  r.w[0] = kind; r.w[1] = tile; r.w[2] = stage; r.w[3] = wave; r.w[4] = piece;
  r.w[5] = (uint32_t)mask; r.w[6] = (uint32_t)(mask >> 32); r.w[7] = got; r.w[8] = exp; r.w[9] = lds_off;
  r.w[10] = reread_ok; r.w[11] = stale_match; r.w[12] = diag_lds_alloc(); r.w[13] = diag_hw_id();
  r.w[14] = diag_xcc_id(); r.w[15] = extra;
  g_diag_rec[i] = r;
}
#endif

template <int MT>
__device__ __forceinline__ void gemm_dma3_body(const GemmArgs& a, const int bx, const int by, const int tile_id) {
  constexpr int M3 = G3<MT>::M, A3_IMAGE = G3<MT>::A_IMAGE, G3_STAGE = G3<MT>::STAGE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int m0 = by * M3, n0 = bx * N3;
  const int M = a.m_dev ? min(a.M, *a.m_dev) : a.M;      // (rows that exist: GemmArgs::m_dev)
  if (m0 >= M) return;
  if (a.gate_tokens) {        // uniform early exit: none of this tile's images needs the map
    // (every wave evaluates the same predicate: no LDS word, the two stages are all of this kernel's LDS)
    const int g0 = m0 / a.gate_rows, g1 = min(m0 + M3 - 1, M - 1) / a.gate_rows;
    const int per = a.gate_T;
    bool need = false;
    for (int i = lane; i < (g1 - g0 + 1) * per; i += 64) {
      const int g = g0 + i / per, t = i % per;
      const int tok = a.gate_tokens[(size_t)t * a.gate_N + g];
      need |= tok >= 0 && tok < a.gate_V && a.gate_token_op[tok] == a.gate_op;
    }
    if (!__any(need)) return;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int nst = a.Kp / K3;

  // ---- DMA roles: wave w moves A rows [8w, 8w + 8) (one piece of 8 rows x 128 B) and pieces 3w .. 3w + 2
  // of the stage's 24 KiB of B planes ---------------------------------------------------------------
  uint32_t a_off[MT], a_sw[MT];
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    const int row = 8 * (w + 8 * u) + (lane >> 3);
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
  const uint16_t* const Bp = a.Bp3;
  const uint32_t b_tile = (uint32_t)bx * (uint32_t)B3_IMAGE + (uint32_t)(3 * w) * 1024u + (uint32_t)lane * 16u;
  const uint32_t b_slab = (uint32_t)(a.Np / N3) * (uint32_t)B3_IMAGE;       // bytes per 32 k of B planes
  const int Klast = a.K - 4;
  auto issue = [&](int s) {
#ifdef N2NMN_DIAG_TWICE
    s = __builtin_amdgcn_readfirstlane(s);       // (inside the pass loop hipcc keeps the stage index in a VGPR)
#endif
    const uint32_t st = lds0 + (uint32_t)(s & 1) * G3_STAGE;
    // columns past K (Kp padding) meet zero weights: any finite in-bounds value will do
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int k = min(s * K3 + 4 * (int)a_sw[u], Klast);
      glds16_3(Ap, a_off[u] + (uint32_t)k * 4u, st + (uint32_t)(8 * (w + 8 * u)) * 128u);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
      glds16_3(Bp, b_tile + (uint32_t)s * b_slab + (uint32_t)u * 1024u,
               st + A3_IMAGE + (uint32_t)(3 * w + u) * 1024u);
  };

#if defined(N2NMN_DIAG_WG) || defined(N2NMN_DIAG_VERIFY) || defined(N2NMN_DIAG_ACC2)
  const uint32_t dlvl = g_diag_level;
#endif
#ifdef N2NMN_DIAG_WG
  if ((dlvl & 8u) && tid == 0 && g_diag_wg) {
    const uint32_t i = atomicAdd(&g_diag_cnt[1], 1u);
    if (i < g_diag_wg_cap) {
      g_diag_wg[4 * i + 0] = diag_lds_alloc(); g_diag_wg[4 * i + 1] = diag_hw_id();
      g_diag_wg[4 * i + 2] = diag_xcc_id(); g_diag_wg[4 * i + 3] = (uint32_t)tile_id;
    }
  }
#endif
#ifdef N2NMN_DIAG_VERIFY
  // where wave w2's piece pc of stage s2 comes from (byte offset from Ap / Bp, per lane) and where it lands
  auto d_src = [&](int w2, int pc, int s2) -> uint32_t {
    if (pc < MT) {
      const int row = 8 * (w2 + 8 * pc) + (lane >> 3);
      int gm = m0 + row;
      gm = gm < M ? gm : M - 1;
      if (a.group_idx) {
        const int g = gm / a.group_size;
        gm = a.group_idx[g] * a.group_size + (gm - g * a.group_size);
      }
      const int sw = (lane & 7) ^ ((row >> 1) & 7);
      const int k = min(s2 * K3 + 4 * sw, Klast);
      return (uint32_t)gm * (uint32_t)a.lda * 4u + (uint32_t)k * 4u;
    }
    return (uint32_t)bx * (uint32_t)B3_IMAGE + (uint32_t)(3 * w2 + (pc - MT)) * 1024u + (uint32_t)lane * 16u +
           (uint32_t)s2 * b_slab;
  };
  auto d_lds = [&](int w2, int pc, int s2) -> uint32_t {
    const uint32_t st = (uint32_t)(s2 & 1) * G3_STAGE + (uint32_t)lane * 16u;
    return pc < MT ? st + (uint32_t)(8 * (w2 + 8 * pc)) * 128u
                   : st + A3_IMAGE + (uint32_t)(3 * w2 + (pc - MT)) * 1024u;
  };
  auto d_verify = [&](uint32_t kind, int w2, int s2) {
    for (int pc = 0; pc < MT + 3; ++pc) {
      const char* gbase = pc < MT ? reinterpret_cast<const char*>(Ap) : reinterpret_cast<const char*>(Bp);
      const uint4 g = *reinterpret_cast<const uint4*>(gbase + d_src(w2, pc, s2));
      const uint32_t lo = d_lds(w2, pc, s2);
      const uint4 l = diag_lds_read(lds0 + lo);
      const bool bad = diag_ne(g, l);
      const uint64_t mask = __ballot(bad);
      if (mask) {
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);
        const uint4 l2 = diag_lds_read(lds0 + lo);
        const uint64_t mask2 = __ballot(diag_ne(g, l2));
        uint32_t stale = 0xffffu;
        if (s2 >= 2) {
          const uint4 g2 = *reinterpret_cast<const uint4*>(gbase + d_src(w2, pc, s2 - 2));
          stale = (uint32_t)__popcll(__ballot(bad && !diag_ne(g2, l)));
        }
        const int fl = __ffsll((long long)mask) - 1;
        const uint32_t got = (uint32_t)__builtin_amdgcn_readlane((int)l.x, fl);
        const uint32_t exp = (uint32_t)__builtin_amdgcn_readlane((int)g.x, fl);
        if (lane == 0)
          diag_record(kind, (uint32_t)tile_id, (uint32_t)s2, (uint32_t)w | ((uint32_t)w2 << 8), (uint32_t)pc, mask,
                      got, exp, lds0 + lo - (uint32_t)lane * 16u, mask2 == 0 ? 1u : 0u, stale,
                      (uint32_t)__popcll(mask) | ((uint32_t)MT << 16));
      }
    }
  };
#endif
#ifdef N2NMN_DIAG_ACC2
  f32x4 acc2[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc2[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif

  // ---- fragment addressing: MFMA 16x16x32 -- lane holds 8 consecutive k (k group lane >> 4) of row /
  // column lane & 15 ----------------------------------------------------------------------------------
  const int li = lane & 15, kg = lane >> 4;
  uint32_t a_c0[MT], a_c1[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int arow = (wm * MT + t) * 16 + li;
    const uint32_t a_x = (uint32_t)((arow >> 1) & 7);
    a_c0[t] = (uint32_t)arow * 128u + ((((uint32_t)(2 * kg)) ^ a_x) << 4);
    a_c1[t] = (uint32_t)arow * 128u + ((((uint32_t)(2 * kg + 1)) ^ a_x) << 4);
  }
  // B planes of this wave's column half: [plane][column tile 4 wn + ct][lane][16 B]
  const uint32_t b_base = (uint32_t)A3_IMAGE + (uint32_t)(4 * wn) * 1024u + (uint32_t)lane * 16u;
  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  struct BFrag { uint4 p[3]; };
  auto read_b = [&](const char* st, int ct) {
    BFrag f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      f.p[p] = *reinterpret_cast<const uint4*>(st + b_base + (uint32_t)(p * 8 + ct) * 1024u);
    return f;
  };

#ifdef N2NMN_DIAG_TWICE
  // the whole K loop a second time over the same operands (the SAME instructions: the pass loop is not
  // unrolled): accumulators that differ between the passes = a transient fault between LDS and the
  // accumulators; equal accumulators with a wrong result = the kernel's inputs differed at the time
  f32x4 accA[MT][4];
  const int npass = (dlvl & 16u) ? 2 : 1;
  uint4 a_sum0 = make_uint4(0u, 0u, 0u, 0u);
  if (dlvl & 32u)
    for (int s2 = 0; s2 < nst; ++s2)
      for (int pc = 0; pc < MT + 3; ++pc) {
        const char* gbase = pc < MT ? reinterpret_cast<const char*>(Ap) : reinterpret_cast<const char*>(Bp);
        const uint4 g = *reinterpret_cast<const uint4*>(gbase + d_src(w, pc, s2));
        a_sum0.x ^= g.x; a_sum0.y += g.y; a_sum0.z ^= g.z; a_sum0.w += g.w;
      }
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    if (pass == 1) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) { accA[t][c] = acc[t][c]; acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      __syncthreads();
    }
#endif
  issue(0);
  for (int s = 0; s < nst; ++s) {
    // this wave's pieces of stage s have landed; after the barrier everyone's have, and everyone is
    // done reading stage s - 1, whose buffer the DMA of stage s + 1 refills under this stage's MFMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef N2NMN_DIAG_VERIFY
    if (dlvl & 1u) d_verify(1u, w, s);
#endif
    __builtin_amdgcn_s_barrier();
#ifdef N2NMN_DIAG_VERIFY
    if (dlvl & 2u) {
      d_verify(2u, (w + 1) & 7, s);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
#endif
    if (s + 1 < nst) issue(s + 1);
    const char* st = smem + (size_t)(s & 1) * G3_STAGE;
    float4 x0[MT], x1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      x0[t] = *reinterpret_cast<const float4*>(st + a_c0[t]);
      x1[t] = *reinterpret_cast<const float4*>(st + a_c1[t]);
    }
    BFrag P = read_b(st, 0);
    // the fp32 A fragments as three bf16 planes (x = hi + mid + lo exactly)
    bf16x8 Ah[MT], Am[MT], Al[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      uint32_t ah[4], am[4], al[4];
      split3(x0[t].x, x0[t].y, ah[0], am[0], al[0]);
      split3(x0[t].z, x0[t].w, ah[1], am[1], al[1]);
      split3(x1[t].x, x1[t].y, ah[2], am[2], al[2]);
      split3(x1[t].z, x1[t].w, ah[3], am[3], al[3]);
      Ah[t] = __builtin_bit_cast(bf16x8, make_uint4(ah[0], ah[1], ah[2], ah[3]));
      Am[t] = __builtin_bit_cast(bf16x8, make_uint4(am[0], am[1], am[2], am[3]));
      Al[t] = __builtin_bit_cast(bf16x8, make_uint4(al[0], al[1], al[2], al[3]));
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      BFrag Q;
      if (ct + 1 < 4) Q = read_b(st, ct + 1);
      __builtin_amdgcn_sched_barrier(0);       // (left alone, hipcc sinks the reads below the MFMAs)
      const bf16x8 Bh = __builtin_bit_cast(bf16x8, P.p[0]), Bm = __builtin_bit_cast(bf16x8, P.p[1]),
                   Bl = __builtin_bit_cast(bf16x8, P.p[2]);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        f32x4 c = acc[t][ct];
        // the six products: small terms first, the leading term last
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[t], Bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[t], Bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[t], Bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[t], Bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[t], Bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[t], Bh, c, 0, 0, 0);
        acc[t][ct] = c;
#ifdef N2NMN_DIAG_ACC2
        if (dlvl & 4u) {
          f32x4 c2 = acc2[t][ct];
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[t], Bh, c2, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[t], Bl, c2, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[t], Bm, c2, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[t], Bh, c2, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[t], Bm, c2, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[t], Bh, c2, 0, 0, 0);
          acc2[t][ct] = c2;
        }
#endif
      }
      if (ct + 1 < 4) P = Q;
    }
  }
#ifdef N2NMN_DIAG_TWICE
  }
  if (npass == 2) {
    int first = -1, nbad = 0;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (__float_as_uint(acc[t][c][r]) != __float_as_uint(accA[t][c][r])) {
            if (first < 0) first = t * 16 + c * 4 + r;
            ++nbad;
          }
    const uint64_t mask = __ballot(nbad > 0);
    if (mask) {
      const int fl = __ffsll((long long)mask) - 1;
      uint32_t got = 0, exp = 0;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (first == t * 16 + c * 4 + r) { got = __float_as_uint(acc[t][c][r]); exp = __float_as_uint(accA[t][c][r]); }
      const uint32_t f1 = (uint32_t)__builtin_amdgcn_readlane(first, fl), n1 = (uint32_t)__builtin_amdgcn_readlane(nbad, fl);
      got = (uint32_t)__builtin_amdgcn_readlane((int)got, fl);
      exp = (uint32_t)__builtin_amdgcn_readlane((int)exp, fl);
      if (lane == 0)
        diag_record(4u, (uint32_t)tile_id, (uint32_t)fl, (uint32_t)w, n1, mask, got, exp, f1, 0u, 0u,
                    (uint32_t)__popcll(mask) | ((uint32_t)MT << 16));
    }
  }
  if (dlvl & 32u) {
    uint4 a_sum1 = make_uint4(0u, 0u, 0u, 0u);
    for (int s2 = 0; s2 < nst; ++s2)
      for (int pc = 0; pc < MT + 3; ++pc) {
        const char* gbase = pc < MT ? reinterpret_cast<const char*>(Ap) : reinterpret_cast<const char*>(Bp);
        const uint4 g = *reinterpret_cast<const uint4*>(gbase + d_src(w, pc, s2));
        a_sum1.x ^= g.x; a_sum1.y += g.y; a_sum1.z ^= g.z; a_sum1.w += g.w;
      }
    const uint64_t mask = __ballot(diag_ne(a_sum0, a_sum1));
    const int M1 = a.m_dev ? min(a.M, *reinterpret_cast<const volatile int*>(a.m_dev)) : a.M;
    if ((mask || M1 != M) && lane == 0)
      diag_record(6u, (uint32_t)tile_id, 0u, (uint32_t)w, 0u, mask, (uint32_t)M1, (uint32_t)M, 0u, 0u, 0u,
                  (uint32_t)__popcll(mask) | ((uint32_t)MT << 16));
  }
#endif

#ifdef N2NMN_DIAG_ACC2
  if (dlvl & 4u) {
    bool bad = false;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= __float_as_uint(acc[t][c][r]) != __float_as_uint(acc2[t][c][r]);
    const uint64_t mask = __ballot(bad);
    if (mask && lane == 0)
      diag_record(3u, (uint32_t)tile_id, 0u, (uint32_t)w, 0u, mask, 0u, 0u, 0u, 0u, 0u, (uint32_t)MT << 16);
  }
#endif
#ifdef N2NMN_DMA3_EPI_LDS
  // ---- epilogue through LDS: the 16x16 MFMA C layout (col = lane & 15, row = 4 (lane >> 4) + r) gives 64-byte
  // row pieces per store instruction; staged as a row-major tile the workgroup writes 16 B per lane, 512 contiguous
  // bytes per row -- whole 128-byte lines, as the fp32 kernel's 32x32 layout does by itself --------------------
  {
    constexpr int LDT = N3 + 4;                                 // row stride of the staged tile (floats)
    float* const tile = reinterpret_cast<float*>(smem);
    __syncthreads();                                            // (every wave is done with the last stage)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tile[((wm * MT + t) * 16 + 4 * kg + r) * LDT + wn * 64 + 16 * ct + li] = acc[t][ct][r];
    __syncthreads();
    for (int idx = tid; idx < M3 * (N3 / 4); idx += G3_THREADS) {
      const int trow = idx / (N3 / 4), c4 = idx - trow * (N3 / 4);
      const int row = m0 + trow, col = n0 + 4 * c4;
      if (row >= M || col >= a.n_store) continue;
      int orow = row;
      if (a.c_row_idx) {
        orow = a.c_row_idx[row];
        if (orow < 0) continue;
      }
      const float4 v = *reinterpret_cast<const float4*>(tile + trow * LDT + 4 * c4);
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cj = col + j;
        float val = cj < a.N ? o[j] + (a.bias ? a.bias[cj] : 0.f) : 0.f;
        if (a.relu) val = fmaxf(val, 0.f);
        o[j] = val;
      }
      float* dst = a.C + (size_t)orow * a.ldc + col;
      if (col + 3 < a.n_store && (a.ldc & 3) == 0) {
        float4 w4 = make_float4(o[0], o[1], o[2], o[3]);
        if (a.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(dst);
          w4.x += old.x; w4.y += old.y; w4.z += old.z; w4.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = w4;
      } else {
        for (int j = 0; j < 4 && col + j < a.n_store; ++j) dst[j] = a.accumulate ? dst[j] + o[j] : o[j];
      }
    }
  }
#else
  // ---- epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + r -----------------
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int col = n0 + wn * 64 + 16 * ct + li;
    if (col >= a.n_store) continue;
    const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * MT + t) * 16 + 4 * kg + r;
        if (row < M) {
          float val = (col < a.N) ? acc[t][ct][r] + bias : 0.f;
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
#endif
#if defined(N2NMN_DIAG_END) && N2NMN_DIAG_END == 1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the wave ends only after its stores were acknowledged
#elif defined(N2NMN_DIAG_END) && N2NMN_DIAG_END == 2
  __threadfence();                                            // agent-scope release: L2 write-back of this XCD
#elif defined(N2NMN_DIAG_END) && N2NMN_DIAG_END == 3
  __builtin_amdgcn_s_sleep(127);                              // (control: the same delay without any memory effect)
  __builtin_amdgcn_s_sleep(127);
#endif
}

#ifndef N2NMN_DIAG_WAVES_PER_EU
#define N2NMN_DIAG_WAVES_PER_EU 4
#endif
template <int MT>
__global__ __launch_bounds__(G3_THREADS, N2NMN_DIAG_WAVES_PER_EU) void gemm_dma3_kernel(GemmBatch b) {
  // consecutive workgroup ids go round-robin over the 8 XCDs: runs of 8 consecutive list positions
  // (column tiles of neighbouring row tiles, which share their A rows) execute on ONE XCD, the runs rotate
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = (((slot >> 3) << 3) + xcd) * 8 + (slot & 7);
  if (t >= b.start[4]) return;
  const int p = (t >= b.start[1]) + (t >= b.start[2]) + (t >= b.start[3]);
  const GemmArgs& a = b.a[p];
  const int local = t - b.start[p];
  const int gx = (a.n_store + N3 - 1) / N3;
  gemm_dma3_body<MT>(a, local % gx, local / gx, t);
}

// PK pack [Kp/4][Np][4] fp32 (zero padded) -> planes [Kp/32][Np/128][3][8][64][8] bf16
__global__ __launch_bounds__(256) void pack_pk_b3_kernel(const float* __restrict__ Bp, int Kp, int Np,
                                                         uint16_t* __restrict__ dst) {
  const size_t total = (size_t)(Kp / 32) * (Np / 128) * 8 * 64;        // fragments per plane
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), ct = (int)((i >> 6) & 7);
    const size_t r = i >> 9;
    const int nt = (int)(r % (Np / 128)), kt = (int)(r / (Np / 128));
    const int n = 128 * nt + 16 * ct + (lane & 15), k0 = 32 * kt + 8 * (lane >> 4);
    uint32_t pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + 2 * e;                                          // k, k + 1: same k4 group
      const float x0 = Bp[((size_t)(k >> 2) * Np + n) * 4 + (k & 3)];
      const float x1 = Bp[((size_t)(k >> 2) * Np + n) * 4 + (k & 3) + 1];
      split3(x0, x1, pl[0][e], pl[1][e], pl[2][e]);
    }
    const size_t base = ((size_t)kt * (Np / 128) + nt) * 3;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint4*>(dst + (((base + p) * 8 + ct) * 64 + lane) * 8) =
          make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
  }
}

}  // namespace

bool gemm_dma3_supported(const GemmArgs& a) {
  const size_t a_rows = a.group_idx ? (size_t)a.src_rows : (size_t)a.M;
  if (!a.Bp3 || (a.group_idx && a.src_rows <= 0)) return false;
  return a.M >= 64 && a.Np % N3 == 0 && a.Kp % K3 == 0 && a.K % 4 == 0 && a.K >= 4 && a.lda % 4 == 0 &&
         a.ksplit <= 1 && a_rows * a.lda * 4 < ((size_t)1 << 32) &&
         (size_t)a.Kp * a.Np * 6 < ((size_t)1 << 32) && (!a.gate_tokens || a.gate_T <= 64);
}

#ifdef N2NMN_DIAG
static int g_diag_lds_pad = 0;       // extra dynamic LDS per workgroup (bytes): changes how workgroups share a CU
#endif
#ifdef N2NMN_DIAG_SUMS
// position-sensitive checksums of what a launch READS (A, gate tokens) and of what it WROTE (C), per problem:
// which launch of a pass differs from the same pass run alone, and whether its inputs already did
namespace {
struct SumRec { const void* C; const void* A; int M, n_store, gated, mt; int slot; };
static std::mutex g_sum_mu;
static std::vector<SumRec> g_sum_recs;
static unsigned long long* g_sum_dev = nullptr;          // [cap][4]: A, tokens, C, -
static const int SUM_CAP = 1 << 14;
__global__ void diag_empty_kernel() {}
__global__ void diag_delay_kernel(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }
__global__ __launch_bounds__(256) void diag_sum_kernel(const uint32_t* p, size_t rows, size_t row_words,
                                                       size_t ld_words, unsigned long long* out) {
  unsigned long long h = 0;
  const size_t total = rows * row_words;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / row_words, c = i - r * row_words;
    h += (unsigned long long)p[r * ld_words + c] * (2ull * i + 1ull);
  }
  atomicAdd(out, h);
}
static void diag_sum(const void* p, size_t rows, size_t row_words, size_t ld_words, unsigned long long* out,
                     hipStream_t s) {
  const size_t total = rows * row_words;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 1024);
  hipLaunchKernelGGL(diag_sum_kernel, dim3(blocks), dim3(256), 0, s, (const uint32_t*)p, rows, row_words, ld_words, out);
}
}  // namespace
#endif
template <int MT>
static void launch3(const GemmArgs* a, int n, hipStream_t s) {
  constexpr int M3 = G3<MT>::M;
  GemmBatch b{};
  int tiles = 0, np = 0;
  for (int i = 0; i < n && np < 4; ++i) {
    if (a[i].M <= 0) continue;
    b.a[np] = a[i];
    b.start[np] = tiles;
    tiles += ((a[i].n_store + N3 - 1) / N3) * ((a[i].M + M3 - 1) / M3);
    ++np;
  }
  if (!np) return;
  for (int i = np; i <= 4; ++i) b.start[i] = tiles;
#ifdef N2NMN_DIAG_SUMS
  // N2NMN_DIAG_SUMS_MODE (diagnostic build only): 1 checksum of the inputs in front of the launch, 2 checksum of
  // the output behind it, 4 / 8 an EMPTY kernel in front / behind instead (which neighbour hides the fault?)
  static const int sums_mode = N2NMN_KNOB_INT("N2NMN_DIAG_SUMS_MODE", 3);
  if (sums_mode & 4) hipLaunchKernelGGL(diag_empty_kernel, dim3(1), dim3(64), 0, s);
  if (sums_mode & 128) hipLaunchKernelGGL(diag_delay_kernel, dim3(1), dim3(64), 0, s, 12);     // ~40 us in FRONT of the launch
  int slots[4] = {-1, -1, -1, -1};
  if (g_sum_dev) {
    std::lock_guard<std::mutex> lk(g_sum_mu);
    for (int i = 0; i < np; ++i) {
      const GemmArgs& g = b.a[i];
      if (g.c_row_idx || g.group_idx || g.m_dev || (int)g_sum_recs.size() >= SUM_CAP) continue;
      slots[i] = (int)g_sum_recs.size();
      g_sum_recs.push_back(SumRec{g.C, g.A, g.M, g.n_store, g.gate_tokens != nullptr, MT, slots[i]});
    }
  }
  for (int i = 0; i < np; ++i)
    if (slots[i] >= 0 && (sums_mode & 1)) {
      const GemmArgs& g = b.a[i];
      diag_sum(g.A, (size_t)g.M, (size_t)g.K, (size_t)g.lda, g_sum_dev + 4 * (size_t)slots[i], s);
      if (g.gate_tokens)
        diag_sum(g.gate_tokens, (size_t)g.gate_T, (size_t)g.gate_N, (size_t)g.gate_N, g_sum_dev + 4 * (size_t)slots[i] + 1, s);
    }
#endif
#ifdef N2NMN_DIAG_LAUNCH
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma3_kernel<MT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G3<MT>::STAGE + g_diag_lds_pad);
  hipLaunchKernelGGL(gemm_dma3_kernel<MT>, dim3((tiles + 63) / 64 * 64), dim3(G3_THREADS),
                     2 * G3<MT>::STAGE + g_diag_lds_pad, s, b);
  return;
#endif
  static std::atomic<uint64_t> attr{0};
  ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_dma3_kernel<MT>), 2 * G3<MT>::STAGE, attr);
  hipLaunchKernelGGL(gemm_dma3_kernel<MT>, dim3((tiles + 63) / 64 * 64), dim3(G3_THREADS), 2 * G3<MT>::STAGE, s, b);
#ifdef N2NMN_DIAG_SUMS
  if (sums_mode & 8) hipLaunchKernelGGL(diag_empty_kernel, dim3(1), dim3(64), 0, s);
  if (sums_mode & 16) hipLaunchKernelGGL(diag_delay_kernel, dim3(1), dim3(64), 0, s, 12);      // ~40 us, no memory access
  if (sums_mode & 64) hipLaunchKernelGGL(diag_delay_kernel, dim3(2048), dim3(256), 0, s, 6);   // ~20 us on every CU
  if (sums_mode & 32)                                  // reads one row in 16 of every output
    for (int i = 0; i < np; ++i)
      if (slots[i] >= 0 && !b.a[i].gate_tokens)
        diag_sum(b.a[i].C, (size_t)b.a[i].M / 16, (size_t)b.a[i].n_store, (size_t)b.a[i].ldc * 16,
                 g_sum_dev + 4 * (size_t)slots[i] + 3, s);
  for (int i = 0; i < np; ++i)
    if (slots[i] >= 0 && (sums_mode & 2) && !b.a[i].gate_tokens)   // (a gated problem leaves the rows of skipped tiles as they were)
      diag_sum(b.a[i].C, (size_t)b.a[i].M, (size_t)b.a[i].n_store, (size_t)b.a[i].ldc,
               g_sum_dev + 4 * (size_t)slots[i] + 2, s);
#endif
}

void launch_gemm_dma3(const GemmArgs* a, int n, hipStream_t s) {
  // 128-row tiles when they still cover the chip a few times over (N2NMN_GEMM_DMA3_MT overrides)
  static const int mt_env = N2NMN_KNOB_INT("N2NMN_GEMM_DMA3_MT", 0);
  int tiles128 = 0;
  for (int i = 0; i < n; ++i)
    if (a[i].M > 0) tiles128 += ((a[i].n_store + N3 - 1) / N3) * ((a[i].M + 127) / 128);
  const int mt = mt_env ? mt_env : (tiles128 >= 1024 ? 2 : 1);
  if (mt == 2) launch3<2>(a, n, s); else launch3<1>(a, n, s);
}

void launch_pack_pk_b3(const float* Bp, int Kp, int Np, uint16_t* dst, hipStream_t s) {
  const size_t total = (size_t)(Kp / 32) * (Np / 128) * 8 * 64;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_pk_b3_kernel, dim3(blocks), dim3(256), 0, s, Bp, Kp, Np, dst);
}

}  // namespace n2nmn

#ifdef N2NMN_DIAG
// ---- diagnostic exports (diag build only; bound by tools/diag/dma3_fault.py with ctypes, not in n2nmn.h) ----
namespace n2nmn {
namespace {
// A co-resident "foreign" workgroup for the concurrency experiments.  mode 0: holds `lds` bytes of LDS and
// sleeps; 1: rewrites and re-reads its own LDS all the time; 2: streams `buf` from HBM (no LDS traffic);
// 3: LDS-DMA from `buf` into its own LDS all the time.
__global__ __launch_bounds__(256) void diag_aggressor_kernel(int mode, int iters, int lds_bytes, const float* buf,
                                                             size_t nfloat, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float acc = 0.f;
  if (lds_bytes >= 4) reinterpret_cast<volatile float*>(dsm)[0] = 1.f;
  if (mode == 0) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
  } else if (mode == 1) {
    const int nw = lds_bytes / 4;
    for (int i = 0; i < iters; ++i) {
      for (int j = tid; j < nw; j += 256) reinterpret_cast<volatile uint32_t*>(dsm)[j] = 0xdead0000u + (uint32_t)i;
      __syncthreads();
      for (int j = tid; j < nw; j += 256) acc += (float)(reinterpret_cast<volatile uint32_t*>(dsm)[j] & 1u);
      __syncthreads();
    }
  } else if (mode == 2) {
    const size_t n4 = nfloat / 4;
    size_t j = ((size_t)blockIdx.x * 256 + tid) % n4;
    for (int i = 0; i < iters; ++i) {
      const float4 v = reinterpret_cast<const float4*>(buf)[j];
      acc += v.x + v.y + v.z + v.w;
      j += (size_t)gridDim.x * 256;
      if (j >= n4) j -= n4;
    }
  } else {
    const uint32_t l0 = (uint32_t)(uintptr_t)dsm;
    const int pieces = lds_bytes / 1024;
    const size_t nb = nfloat * 4;
    size_t off = ((size_t)blockIdx.x * 4 + w) * 1024 % (nb - 2048);
    for (int i = 0; i < iters; ++i) {
      for (int pc = w; pc < pieces; pc += 4) {
        glds16_3(buf, (uint32_t)(off + (size_t)lane * 16),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + (uint32_t)pc * 1024u)));
        off += (size_t)gridDim.x * 4096;
        if (off >= nb - 2048) off -= nb - 2048;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    acc += reinterpret_cast<volatile float*>(dsm)[tid];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
}  // namespace
}  // namespace n2nmn

extern "C" {
int n2nmn_diag_config(unsigned level, void* rec, unsigned rec_cap, void* cnt, void* wg, unsigned wg_cap,
                      int lds_pad) {
  using namespace n2nmn;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag_level), &level, sizeof(level));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag_rec), &rec, sizeof(rec));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag_cap), &rec_cap, sizeof(rec_cap));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag_cnt), &cnt, sizeof(cnt));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag_wg), &wg, sizeof(wg));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag_wg_cap), &wg_cap, sizeof(wg_cap));
  g_diag_lds_pad = lds_pad;
  return e == hipSuccess ? 0 : -(int)e;
}
// PK pack [Kp/4][Np][4] of a row-major B[K][N], then its bf16 planes
int n2nmn_diag_pack3(const float* B, int K, int N, float* Bp, void* Bp3, int Kp, int Np, void* stream) {
  n2nmn::launch_pack_pk(B, N, K, N, Bp, Kp, Np, (hipStream_t)stream);
  n2nmn::launch_pack_pk_b3(Bp, Kp, Np, (uint16_t*)Bp3, (hipStream_t)stream);
  return 0;
}
// one launch, no synchronisation; mt = 1 / 2: gemm_dma3_kernel<mt>, mt = 0: the exact-fp32 gemm_dma_kernel
int n2nmn_diag_gemm(const float* A, const float* Bp, const void* Bp3, const float* bias, float* C, int M, int N,
                    int K, int Kp, int Np, int mt, void* stream) {
  using namespace n2nmn;
  GemmArgs g{};
  g.A = A; g.lda = K; g.M = M; g.K = K; g.group_size = 1; g.Bp = Bp; g.Np = Np; g.Kp = Kp;
  g.Bp3 = (const uint16_t*)Bp3; g.bias = bias; g.N = N; g.C = C; g.ldc = N; g.n_store = N;
  if (mt == 0) { if (!gemm_dma_supported(g)) return -1; launch_gemm_dma(&g, 1, (hipStream_t)stream); }
  else { if (!gemm_dma3_supported(g)) return -1; if (mt == 2) launch3<2>(&g, 1, (hipStream_t)stream); else launch3<1>(&g, 1, (hipStream_t)stream); }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
#ifdef N2NMN_DIAG_SUMS
// start / read the per-launch checksum records: out[i] = {C pointer, A pointer, M | n_store << 32, gated | mt << 8,
// sum(A), sum(tokens), sum(C), 0}; returns the number of records
int n2nmn_diag_sums_reset() {
  using namespace n2nmn;
  std::lock_guard<std::mutex> lk(g_sum_mu);
  if (!g_sum_dev && hipMalloc(reinterpret_cast<void**>(&g_sum_dev), sizeof(unsigned long long) * 4 * SUM_CAP) != hipSuccess)
    return -1;
  (void)hipDeviceSynchronize();
  (void)hipMemset(g_sum_dev, 0, sizeof(unsigned long long) * 4 * SUM_CAP);
  (void)hipDeviceSynchronize();
  g_sum_recs.clear();
  return 0;
}
int n2nmn_diag_sums_get(unsigned long long* out, int cap) {
  using namespace n2nmn;
  std::lock_guard<std::mutex> lk(g_sum_mu);
  (void)hipDeviceSynchronize();
  const int n = std::min<int>((int)g_sum_recs.size(), cap);
  std::vector<unsigned long long> dev(4 * (size_t)std::max(n, 1));
  if (n) (void)hipMemcpy(dev.data(), g_sum_dev, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) {
    const SumRec& r = g_sum_recs[i];
    out[8 * i + 0] = (unsigned long long)(uintptr_t)r.C; out[8 * i + 1] = (unsigned long long)(uintptr_t)r.A;
    out[8 * i + 2] = (unsigned long long)(unsigned)r.M | ((unsigned long long)(unsigned)r.n_store << 32);
    out[8 * i + 3] = (unsigned long long)r.gated | ((unsigned long long)r.mt << 8);
    out[8 * i + 4] = dev[4 * i]; out[8 * i + 5] = dev[4 * i + 1]; out[8 * i + 6] = dev[4 * i + 2]; out[8 * i + 7] = 0;
  }
  return n;
}
#endif
int n2nmn_diag_aggressor(int mode, int iters, int lds_bytes, int grid, const float* buf, size_t nfloat, float* sink,
                         void* stream) {
  using namespace n2nmn;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(diag_aggressor_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(diag_aggressor_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, mode, iters,
                     lds_bytes, buf, nfloat, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}  // extern "C"
#endif

