// Backward-pass kernels of the training step (exp_clevr/train_clevr_gt_layout.py:104-130):
// gradients of the behavioural-cloning objective through the layout generator
// (models_clevr/nmn3_netgen_att.py) -- the module-network backward lives in
// kernels_train_modules.hip.  TensorFlow computes these with its registered op gradients; here
// each is a hand-written gfx950 kernel, and every gradient writer ACCUMULATES into the zeroed flat
// gradient buffer.
//
//   gemm_tn        : all weight gradients  dW = X^T . dY          (MFMA 32x32x2 fp32, split-R)
//   colsum         : bias gradients        db = sum_r dY[r, :]
//   lstm_bwd_step  : one reverse-time step of an LSTM layer: dz_{t+1} . W^T on MFMA with the cell
//                    backward of step t fused in the epilogue (BPTT; two layers pipelined per launch)
//   (the gradient of the layer-0 input-projection tables is a one-hot gemm_tn)
//   dec_bwd_a/b    : token-logit loss, additive attention, masked softmax backward
//   word_vecs_bwd  : gradient of the text-attention word vectors
//   loss, grad_finish, grad_sqnorm, adam : objective and optimiser (clip_by_norm + Adam)
#include <algorithm>
#include <type_traits>

#include "device_utils.h"
#include <cstdlib>

#include "kernels.h"

namespace n2nmn {

namespace {

// ---------------------------------------------------------------------------------------------
// gemm_tn: C[m][n] += sum_r A[row(r)][m] * B[r][n]
// Workgroup tile (64 WT) x (64 WT), 4 waves (2x2), each wave WT x WT accumulators of one 32x32x2
// fp32 MFMA; the reduction runs over rows r in k-tiles of BK (64x64x32 or 128x128x16).  Both
// operands arrive with r as the slow axis and stay that way in LDS ([r][m], unpadded): a loading
// thread copies four 16-B pieces per tile global -> registers -> ds_write_b128 with no arithmetic
// in between (lanes along m: coalesced loads, contiguous conflict-free writes), and the MFMA
// operands A[k][m0 + lane % 32] are ds_read_b32 of 32 consecutive words.  Waves 0,1 stream A,
// waves 2,3 stream B (wave-uniform roles: every mode test below is a scalar branch).
//
// The first measured version of this kernel issued 14 VALU instructions per MFMA (per-element
// masks, a register transpose for a k-interleaved LDS layout, 64-bit addresses) and ran at the
// VALU's pace, 37 % of the MFMA peak; tools/gemm_tn_microbench.py, profiles/r01_gemm_tn.txt.
// ---------------------------------------------------------------------------------------------
template <int WT, int BK>
__global__ __launch_bounds__(256, WT == 1 ? 4 : 2) void gemm_tn_kernel(const GemmTnArgs a,
                                                                       int r_per_split) {
  constexpr int TB = 64 * WT;              // tile edge (both M and N)
  constexpr int C4N = TB / 4;              // float4 pieces per tile row
  constexpr int RP = 128 / C4N;            // tile rows covered by one pass of 128 loading threads
  static_assert(BK == 4 * RP, "four pieces per loading thread and tile");
  __shared__ float4 As[2][BK][C4N];
  __shared__ float4 Bs[2][BK][C4N];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
  int Rtot = a.R;
  // blockIdx.z = split * nprob + problem
  const int nprob = a.nprob > 0 ? a.nprob : 1;
  const int bz = blockIdx.z + a.z_off;
  const int zp = bz % nprob, zsplit = bz / nprob, nsplit = a.z_total / nprob;
  // (the argument block is never written: a modified by-value struct is copied to scratch memory)
  const bool multi = a.nprob > 0;
  const float* const Ain = multi ? a.A_p[zp] : a.A;
  const float* const Bin = multi ? a.B_p[zp] : a.B;
  const int b_sel_val = multi ? a.bsel_p[zp] : a.b_sel_val;
  float* const Cout = multi ? a.C_p[zp] : a.C;
  float* const colsum_out = multi ? a.colsum_p[zp] : a.colsum;
  if (a.r_dev) {                       // compacted reduction: re-balance the splits on the device
    Rtot = min(a.R, *a.r_dev);
    const int nkt_all = (Rtot + BK - 1) / BK;
    r_per_split = ((nkt_all + nsplit - 1) / nsplit) * BK;
  }
  const int rbeg = zsplit * r_per_split;
  const int rend = min(Rtot, rbeg + r_per_split);
  if (rbeg >= rend) return;
  const int nkt = (rend - rbeg + BK - 1) / BK;

  const bool isB = wid >= 2;
  const int lt = tid & 127;
  const int trow = lt / C4N;         // tile rows trow + j * RP, j = 0..3
  const int c4 = lt % C4N;           // which 16-B piece of the row (4 consecutive m or n)
  const unsigned ld = isB ? a.ldb : a.lda;
  const int cbeg = (isB ? n0 : m0) + 4 * c4;
  const int clim = isB ? a.N : a.M;
  // Columns >= clim of a tile only reach rows / columns of C that are never written, so operand
  // columns need no mask: a piece starting outside the (4-padded) row just re-reads column 0.
  const unsigned ccl = cbeg < ((clim + 3) & ~3) ? cbeg : 0;

  // Row metadata is a chain of up to two dependent loads (row_idx -> a_group_idx / a_onehot /
  // b_sel) in front of the operand load.  The three stages run as a software pipeline, one k-tile
  // apart, so an iteration issues its loads back to back and waits for none before its MFMAs:
  //   S1(t): l1[j]   = row_idx[r]                      (tile t)
  //   S2(t): l2[j]   = table2[src1 or src1 / group]    (tile t; src1 from l1)
  //   S3(t): data[j] = base[src * ld + col]            (tile t; src / mask from l2)
  const bool use1 = a.row_idx != nullptr;
  const bool onehot = !isB && a.a_onehot != nullptr;
  const bool group = !isB && !onehot && a.a_group_idx != nullptr;
  const bool bsel = isB && a.b_sel != nullptr;
  const bool use2 = onehot || group || bsel;
  const int32_t* const p2 = onehot ? a.a_onehot : group ? a.a_group_idx : a.b_sel;
  const int gs = group ? a.a_group_size : 1;
  const float inv_gs = 1.0f / (float)gs;
  const float* const base = isB || onehot ? Bin : Ain;     // (one-hot: any readable address)

  // bias gradient riding along: the workgroups of the first row of output tiles also sum the B
  // columns they stream (every B tile is loaded exactly once per workgroup)
  const bool do_cs = isB && colsum_out != nullptr && blockIdx.y == 0;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  int l1[4], l2[4], src1[4], rem[4];
  // two operand tiles in flight per thread (slot = tile & 1)
  int aux[2][4];                     // one-hot: the hot column; b_sel: the row's selection value
  float4 reg[2][4];
  auto row_of = [&](int kt, int j) {         // tile row -> reduction row, clamped into the split
    const int r = rbeg + kt * BK + trow + j * RP;
    return r < rend ? r : rbeg;
  };
  auto stage1 = [&](int kt) {
    if (!use1) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) l1[j] = a.row_idx[row_of(kt, j)];
  };
  auto stage2 = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s1 = use1 ? l1[j] : row_of(kt, j);
      src1[j] = s1;
      if (use2) {
        int g = s1;
        if (gs > 1) {                    // exact floor(s1 / gs) for s1 < 2^22
          g = __float2int_rz(((float)s1 + 0.5f) * inv_gs);
          if (g * gs > s1) --g;
          if ((g + 1) * gs <= s1) ++g;
          rem[j] = s1 - g * gs;
        }
        l2[j] = p2[g];
      }
    }
  };
  auto stage3 = [&](auto slot_c, int kt) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned src = src1[j];
      if (group) src = gs > 1 ? l2[j] * gs + rem[j] : l2[j];
      if (use2) aux[SL][j] = l2[j];
      const unsigned off = onehot ? 0u : src * ld + ccl;      // operands span < 2^30 floats
      reg[SL][j] = *reinterpret_cast<const float4*>(base + (size_t)off);
    }
  };
  auto lstore = [&](auto slot_c, int kt) {   // tile kt: register slot kt & 1 -> LDS buffer kt & 1
    constexpr int SL = decltype(slot_c)::value;
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = reg[SL][j];
    if (onehot) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int h = aux[SL][j] - cbeg;
        v[j] = make_float4(h == 0 ? 1.f : 0.f, h == 1 ? 1.f : 0.f, h == 2 ? 1.f : 0.f,
                           h == 3 ? 1.f : 0.f);
      }
    }
    // rows outside the split (last tile) or of another selection contribute zero: B is enough
    if (isB && (bsel || rbeg + (kt + 1) * BK > rend)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bool ok = rbeg + kt * BK + trow + j * RP < rend;
        if (bsel) ok = ok && aux[SL][j] == b_sel_val;
        if (!ok) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (do_cs) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { cs.x += v[j].x; cs.y += v[j].y; cs.z += v[j].z; cs.w += v[j].w; }
    }
    float4(*dst)[C4N] = isB ? Bs[SL] : As[SL];
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[trow + j * RP][c4] = v[j];
  };

  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, kh = lane >> 5;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  stage1(0);
  stage2(0);
  stage1(1);
  stage3(S0{}, 0);
  stage2(1);
  stage1(2);
  stage3(S1{}, 1);
  stage2(2);
  stage1(3);
  lstore(S0{}, 0);
  __syncthreads();
  // iteration kt (parity P): issue tile kt+2's operand loads and the metadata of tiles kt+3 / kt+4
  // (tiles past the end re-read row rbeg, masked), MFMAs of tile kt from LDS buffer P, then tile
  // kt+1 (landed during the previous iteration's MFMAs) moves from its registers to buffer 1-P
  auto body = [&](auto par_c, int kt) {
    constexpr int P = decltype(par_c)::value;
    stage3(par_c, kt + 2);
    stage2(kt + 3);
    stage1(kt + 4);
    const float* Af = reinterpret_cast<const float*>(&As[P][0][0]) + kh * TB + wm * (32 * WT) + li;
    const float* Bf = reinterpret_cast<const float*>(&Bs[P][0][0]) + kh * TB + wn * (32 * WT) + li;
    // operand reads run one group of GS k-pair steps (>= 4 MFMAs) ahead of the MFMAs that use
    // them; left to itself the compiler reloads the same registers after every group and waits
    constexpr int GS = WT == 1 ? 4 : 1, NG = BK / 2 / GS;
    float av[2][GS][WT], bv[2][GS][WT];
    auto lds_group = [&](int set, int g) {
#pragma unroll
      for (int q = 0; q < GS; ++q)
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          av[set][q][i] = Af[(g * GS + q) * 2 * TB + i * 32];
          bv[set][q][i] = Bf[(g * GS + q) * 2 * TB + i * 32];
        }
    };
    lds_group(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) lds_group((g + 1) & 1, g + 1);
      __builtin_amdgcn_sched_barrier(0);       // keep the reads of group g+1 ahead of these MFMAs
#pragma unroll
      for (int q = 0; q < GS; ++q)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][q][i], bv[g & 1][q][j],
                                                             acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nkt) lstore(std::integral_constant<int, 1 - P>{}, kt + 1);
    __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    body(S0{}, kt);
    if (kt + 1 < nkt) body(S1{}, kt + 1);
  }
  // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    const int col = n0 + (wn * WT + j) * 32 + li;
    if (col >= a.N) continue;
#pragma unroll
    for (int i = 0; i < WT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * WT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < a.M) {
          float* dst = Cout + (size_t)row * a.ldc + col;
          if (nsplit == 1) *dst += acc[i][j][r];       // this workgroup owns the tile
          else atomicAdd(dst, acc[i][j][r]);
        }
      }
    }
  }
  if (colsum_out != nullptr && blockIdx.y == 0) {   // uniform per workgroup; the k-loop ended
    float* red = reinterpret_cast<float*>(&Bs[0][0][0]);     // with a barrier.  [RP rows][TB cols]
    if (isB) *reinterpret_cast<float4*>(red + trow * TB + 4 * c4) = cs;
    __syncthreads();
    if (tid < TB && n0 + tid < a.N) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RP; ++q) t += red[q * TB + tid];
      atomicAdd(colsum_out + n0 + tid, t);
    }
  }
}

__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRanges z) {
  const size_t gtid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t gsz = gridDim.x * (size_t)blockDim.x;
  for (int i = 0; i < z.n; ++i) {
    char* p = static_cast<char*>(z.ptr[i]);
    const size_t nb = z.bytes[i];
    // head up to a 16-B boundary, 16-B body, 4-B tail
    const size_t head = min(nb, (size_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15));
    const size_t body = (nb - head) / 16;
    const size_t tail0 = head + body * 16;
    if (gtid < head / 4) reinterpret_cast<float*>(p)[gtid] = 0.f;
    float4* b4 = reinterpret_cast<float4*>(p + head);
    for (size_t j = gtid; j < body; j += gsz) b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gtid < (nb - tail0) / 4) reinterpret_cast<float*>(p + tail0)[gtid] = 0.f;
  }
}

// rows (t, n) inside the question's length, as one compacted list and -- for the weight-gradient
// GEMMs that follow the reverse-time recurrence chunk by chunk -- as one list per time chunk:
// chunk ci covers t in [st[ci], st[ci-1]) (st descending, st[-1] = T, st[3] = 0), its list starts at
// rows_ch + st[ci]*N and its length is count[1 + ci]
__global__ void active_rows_kernel(const int32_t* __restrict__ seq_len, int T, int N,
                                   int32_t* __restrict__ rows, int32_t* __restrict__ count,
                                   int32_t* __restrict__ rows_ch, int4 st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < T * N) {
    const int t = i / N, n = i - t * N;
    if (t < seq_len[n]) {
      rows[atomicAdd(count, 1)] = i;
      if (rows_ch) {
        const int ci = t >= st.x ? 0 : t >= st.y ? 1 : t >= st.z ? 2 : 3;
        const int beg = ci == 0 ? st.x : ci == 1 ? st.y : ci == 2 ? st.z : st.w;
        rows_ch[(size_t)beg * N + atomicAdd(count + 1 + ci, 1)] = i;
      }
    }
  }
}

// dst[c] += sum_r src[r*ld + c]; lanes own consecutive columns, waves / blockIdx.y split the rows
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, int R,
                                                     int ncols, int ld,
                                                     const int32_t* __restrict__ sel, int sel_val,
                                                     float* __restrict__ dst, int r_per_block) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * r_per_block, r1 = min(R, r0 + r_per_block);
  float s0 = 0.f, s1 = 0.f;
  if (c < ncols) {
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) {
      const bool k0 = !sel || sel[r] == sel_val, k1 = !sel || sel[r + 4] == sel_val;
      const float v0 = src[(size_t)r * ld + c], v1 = src[(size_t)(r + 4) * ld + c];
      s0 += k0 ? v0 : 0.f;
      s1 += k1 ? v1 : 0.f;
    }
    if (r < r1 && (!sel || sel[r] == sel_val)) s0 += src[(size_t)r * ld + c];
  }
  part[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && c < ncols)
    atomicAdd(dst + c, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}

__global__ void pack_pk_t_kernel(const float* __restrict__ src, int ld, int K, int N,
                                 float* __restrict__ dst, int Kp, int Np) {
  const size_t total = (size_t)Kp * Np;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 3);
    const size_t r = i >> 2;
    const int n = (int)(r % Np);
    const int k = (int)(r / Np) * 4 + kk;
    dst[i] = (k < K && n < N) ? src[(size_t)n * ld + k] : 0.f;
  }
}

// dst[((j*(Ktot/4) + k_off/4 + u)*16 + c)*4 + g] = W[(row0 + 16j + c)*ld + g*L + u]
__global__ void pack_tiles_t_kernel(const float* __restrict__ W, int ld, int row0, int L,
                                    float* __restrict__ dst, int Ktot, int k_off) {
  const size_t total = (size_t)(L / 16) * L * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i & 3);
    const int c = (int)((i >> 2) & 15);
    const size_t r = i >> 6;
    const int u = (int)(r % L);
    const int j = (int)(r / L);
    dst[(((size_t)j * (Ktot / 4) + k_off / 4 + u) * 16 + c) * 4 + g] =
        W[(size_t)(row0 + 16 * j + c) * ld + (size_t)g * L + u];
  }
}

// ---------------------------------------------------------------------------------------------
// lstm_bwd_step_kernel.  Reverse-time recurrence of one layer (grid.y = job):
//
//   rec[n, u]  = sum_k dz_next[n, k] * Wt[k, u]          k = 4*unit + gate over K = 4L or 8L
//   dh         = rec + (step t+1 masked ? dH[n,u] : 0) + dout[n,u]
//   tc = tanh(c_t);  dc = dC + dh*o*(1-tc^2)
//   dz_i = dc*j*i(1-i)  dz_j = dc*i*(1-j^2)  dz_f = dc*c_{t-1}*f(1-f)  dz_o = dh*tc*o(1-o)
//   dC <- dc*f
//
// (gates saved after their nonlinearities; f already contains the +1 forget bias).  A row past
// its length at step t (dynamic_rnn carries the state through, Appendix A.2) emits dz = 0 and
// leaves dH / dC untouched; "step T" counts as masked for every row so the initial carry (the
// decoder's gradient of the encoder state) enters through dH / dC.
//
// CDNA4 mapping mirrors lstm_step_kernel: a workgroup owns 16 hidden units (one 16x16x4 MFMA
// N-tile) x 16 batch rows; its 8 waves split K and stream dz (k-interleaved [unit][row][4 gates],
// so an A fragment is 4 x 256-B contiguous) and the pre-transposed weight tile straight from L2
// into MFMA registers; partial tiles are reduced through LDS and the cell backward runs in the
// same kernel.  The step is fp32-MFMA bound per workgroup (2*16*16*K flops on one CU), so the row
// block is a single 16-row M tile: 2 jobs x 32 column tiles x N/16 row blocks = 256 workgroups.  Layer 0's job contracts over [dz1_{t}; dz0_{t+1}] (K = 8L) so that the gradient
// from the layer above and the recurrent gradient come out of ONE accumulation.
// ---------------------------------------------------------------------------------------------
// BW_WAVES = waves that split K (template parameter).  A step moves 96 MB from L2 to the CUs at
// lstm_dim 512 / 64 rows (every workgroup streams its 16 rows x K of dz and its K x 16 weight tile: 4
// flops per byte), which is what bounds it -- not the bytes in flight: 16 waves (256 KiB in flight per
// CU) measured 12.3 us per step against 11.7 for 8 (round 3; the 16-wave form stays behind the diagnostic build's N2NMN_BWD_WAVES).
struct LstmBwdJobs {
  LstmBwdJob j[2];
};

template <int BW_MT, int BW_WAVES>
__global__ __launch_bounds__(BW_WAVES * 64) void lstm_bwd_step_kernel(LstmBwdJobs jobs, int N,
                                                                      int L) {
  const LstmBwdJob& jb = jobs.j[blockIdx.y];
  if (!jb.active) return;
  constexpr int BW_THREADS = BW_WAVES * 64;
  __shared__ float part[BW_WAVES][16 * BW_MT][17];
  constexpr int ROWS = 16 * BW_MT;
  const int tile = blockIdx.x;
  const int row0 = blockIdx.z * ROWS;
  // (the active-row count of step t is a dependent scalar load: requested here, tested only after the epilogue
  // operands and the first operand sets have been requested, so its round trip runs under theirs)
  const int32_t* const nact_p = jb.n_act;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = lane & 15, kg = lane >> 4;

  // The cell backward's operands (thread = (row, unit) of the tile) are requested BEFORE the
  // contraction: they were written by the forward pass long ago and come from HBM, a latency the
  // k loop covers instead of the epilogue paying it after the barrier.
  const int erow = tid >> 4, ul = tid & 15;
  const int es = row0 + erow;                    // GEMM row (slot); en = the question in it
  static_assert(16 * 16 * BW_MT <= BW_THREADS, "one thread per (row, unit) of the tile");
  const bool epi = erow < ROWS && es < N;
  const int en = epi ? (jb.perm ? jb.perm[es] : es) : 0;
  const int eu = 16 * tile + ul;
  const size_t idx = (size_t)en * L + eu;
  const bool two_src = jb.drop && jb.A1;
  float4 e_g = make_float4(0.f, 0.f, 0.f, 0.f);
  float e_cn = 0.f, e_cp = 0.f, e_dC = 0.f, e_dH = 0.f, e_dout = 0.f, e_drop = 1.f;
  int e_len = jb.T;
  if (epi) {
    if (jb.seq_len) e_len = jb.seq_len[en];
    e_dH = jb.dH[idx];
    if (two_src) e_drop = jb.drop[idx];
    if (jb.cell) {
      e_g = jb.gates[idx];
      e_cn = jb.c_new[idx];
      e_cp = jb.c_prev[idx];
      e_dC = jb.dC[idx];
      if (jb.dout) e_dout = jb.dout[idx];
    }
  }

  f32x4 acc[BW_MT];
#pragma unroll
  for (int m = 0; m < BW_MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (jb.gemm) {
    const int K = jb.K, R = jb.R;
    const int wu = __builtin_amdgcn_readfirstlane(w);     // (wave-uniform: the bases below live in scalar registers)
    const int kper = K / BW_WAVES;                 // k's of this wave (multiple of 16)
    const int kbeg = wu * kper;
    const int K1 = 4 * L;                          // k's held by A0
    const float* Asrc = (kbeg < K1) ? jb.A0 : jb.A1;
    const int kloc = (kbeg < K1) ? kbeg : kbeg - K1;
    // address = wave-uniform base (advanced per chunk with scalar adds) + a 32-bit lane offset that never changes:
    // chunk q of the weight tile is 64 float4 further, chunk q of dz 4 R float4 further
    const char* const Wb = reinterpret_cast<const char*>(jb.Wt) +
                           ((size_t)tile * (K / 4) * 16 + (size_t)(kbeg >> 2) * 16) * sizeof(float4);
    const uint32_t wl = (uint32_t)(kg * 16 + ci) * (uint32_t)sizeof(float4);
    const char* const Ab = reinterpret_cast<const char*>(Asrc) + (size_t)(kloc >> 2) * R * sizeof(float4);
    const size_t a_step = (size_t)4 * R * sizeof(float4);
    uint32_t al[BW_MT];
#pragma unroll
    for (int m = 0; m < BW_MT; ++m) {
      int r = row0 + 16 * m + ci;
      r = r < N ? r : N - 1;
      al[m] = (uint32_t)(kg * R + r) * (uint32_t)sizeof(float4);
    }
    const int nch = kper / 16;
    // A ring of NS register sets of UN chunks: the loads of set s + NS - 1 are issued BEFORE the MFMAs of set s, so the
    // wave streams its k slice with (NS - 1) UN .. NS UN chunks in flight and no memory round trip between sets
    // (round 6; the single-set loop it replaces kept 8 chunks in flight and paid the L2 latency once per set: 4 times
    // on layer 0, twice on layer 1).  Chunks are accumulated in the same order as before.  Sweep: profiles/r06_notes.md.
#ifndef BWS_UN
#define BWS_UN 1
#endif
#ifndef BWS_NS
#define BWS_NS 4
#endif
    constexpr int UN = BW_MT == 1 ? BWS_UN : (BWS_UN + 1) / 2;   // chunks per set (NS x UN x (1 + BW_MT) float4 registers)
    constexpr int NS = BWS_NS;
    f32x4 bq[NS][UN];
    f32x4 aq[NS][UN][BW_MT];
    // (plain loads: non-temporal ones cost 0.34 ms per step -- the four row blocks of a column tile share its lines)
    auto ld = [&](const char* ptr) -> f32x4 { return *reinterpret_cast<const f32x4*>(ptr); };
    auto load_set = [&](int s, int q0) {           // (s is a compile-time constant wherever this is called)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int q = q0 + u < nch ? q0 + u : nch - 1;
        bq[s][u] = ld(Wb + (size_t)q * (64 * sizeof(float4)) + wl);
#pragma unroll
        for (int m = 0; m < BW_MT; ++m) aq[s][u][m] = ld(Ab + (size_t)q * a_step + al[m]);
      }
    };
    auto mfma_set = [&](int s, int nq) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (u < nq) {
#pragma unroll
          for (int m = 0; m < BW_MT; ++m) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s][u][m][e], bq[s][u][e], acc[m], 0, 0, 0);
          }
        }
      }
    };
    if (NS > 1 && nch % (NS * UN) == 0) {
      // whole rings (lstm_dim a multiple of 32 NS UN): straight-line body, no test per chunk; the scheduling barriers
      // keep the compiler from sinking a set's loads below the MFMAs they are meant to run under
#pragma unroll
      for (int s = 0; s < NS - 1; ++s) load_set(s, s * UN);
      __builtin_amdgcn_sched_barrier(0);
      if (nact_p && row0 >= *nact_p) return;       // no row of this block is inside its length at step t
      int q0 = 0;
      for (; q0 + NS * UN < nch; q0 += NS * UN) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          load_set((s + NS - 1) % NS, q0 + (s + NS - 1) * UN);
          __builtin_amdgcn_sched_barrier(0);
          mfma_set(s, UN);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      load_set(NS - 1, q0 + (NS - 1) * UN);        // the last ring: only its last set is still to be requested
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NS; ++s) mfma_set(s, UN);
    } else {
      if (nact_p && row0 >= *nact_p) return;
      for (int q0 = 0; q0 < nch; q0 += UN) {       // any other size: one set at a time
        load_set(0, q0);
        mfma_set(0, nch - q0);
      }
    }
  }
  else if (nact_p && row0 >= *nact_p) return;
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int m = 0; m < BW_MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][16 * m + 4 * kg + r][ci] = acc[m][r];
  __syncthreads();

  // ---- epilogue: thread = (row, unit) ---------------------------------------------------------
  if (!epi) return;                            // 16*ROWS threads own one (row, unit) each
  const int n = en, u = eu;
  float rec = 0.f;
  if (two_src) {
    // waves [0, BW_WAVES/2) contracted A0 (the gradient from the layer above, which saw this
    // layer's output through the dropout multipliers), the rest A1 (this layer's own recurrence)
    float up = 0.f;
#pragma unroll
    for (int ww = 0; ww < BW_WAVES / 2; ++ww) up += part[ww][erow][ul];
#pragma unroll
    for (int ww = BW_WAVES / 2; ww < BW_WAVES; ++ww) rec += part[ww][erow][ul];
    rec += up * e_drop;
  } else {
#pragma unroll
    for (int ww = 0; ww < BW_WAVES; ++ww) rec += part[ww][erow][ul];
  }
  const int len = e_len;
  const bool m_next = jb.t + 1 >= len;            // step t+1 carried the state through
  const float dh_state = rec + (m_next ? e_dH : 0.f);
  if (!jb.cell) {                                 // gradient of the initial hidden state
    jb.dH[idx] = dh_state;                        // (there is no step -1: its dz operand is zero)
    *reinterpret_cast<float4*>(jb.dz_k + ((size_t)u * jb.R + es) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
  if (jb.t < len) {
    const float dh = dh_state + e_dout;
    const float4 g = e_g;                         // i, j, f, o
    const float tc = fast_tanh(e_cn);
    const float dc = e_dC + dh * g.w * (1.f - tc * tc);
    dz.x = dc * g.y * g.x * (1.f - g.x);
    dz.y = dc * g.x * (1.f - g.y * g.y);
    dz.z = dc * e_cp * g.z * (1.f - g.z);
    dz.w = dh * tc * g.w * (1.f - g.w);
    jb.dC[idx] = dc * g.z;
  }                                               // masked: dH / dC keep carrying
  *reinterpret_cast<float4*>(jb.dz_k + ((size_t)u * jb.R + es) * 4) = dz;
  float* zr = jb.dz_rm + (size_t)n * 4 * L + u;
  zr[0] = dz.x; zr[L] = dz.y; zr[2 * L] = dz.z; zr[3 * L] = dz.w;
}

__global__ void dec_xidx_kernel(const int32_t* __restrict__ gt, int Td, int N, int go_row,
                                int32_t* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Td * N) idx[i] = i < N ? go_row : gt[i - N];
}

// ---------------------------------------------------------------------------------------------
// dec_bwd_a: one workgroup per (question n, decoder step t).  Backward of everything after the
// LSTM cell of that step (nmn3_netgen_att.py:184-256 with use_gt_layout: all tokens valid):
//   dsc   = (softmax(token_scores) - onehot(gt)) / N              d(-mean log_seq_prob)
//   dout  = W_y[:L] . dsc ;  dctx = W_y[L:] . dsc
//   datt  = dctx . eout[tau] + datts_wv ;  de = att * (datt - sum att*datt)   (masked softmax)
//   dq_k  = sum_tau de[tau] v_k (1 - th^2) ;  dv_k partial = sum_tau de[tau] th,
//   th = tanh(q_k + eht[tau,n,k])
// ---------------------------------------------------------------------------------------------
constexpr int DB_MAXKI = 4;     // lstm_dim <= 1024
__global__ __launch_bounds__(256) void dec_bwd_a_kernel(DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, T = a.T, N = a.N, V = a.V;
  float* dctx = smem;                 // [L]
  float* des = dctx + L;              // [T]  datt -> de
  float* ats = des + ((T + 3) & ~3);  // [T]
  float* dscs = ats + ((T + 3) & ~3); // [16]
  float* red = dscs + 16;             // [4 waves][2][L] partial dq / dv
  const int n = blockIdx.x, t = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t tn = (size_t)t * N + n;
  const int len = min(max(a.seq_len[n], 0), a.T);

  if (w == 0) {                       // d token logits
    // p = softmax restricted to the valid tokens (nmn3_netgen_att.py:245-247);
    //   d log p[chosen] / d sc_v = [v == chosen] - p_v            (valid v; invalid ones: 0)
    //   d neg_entropy   / d sc_v = p_v (g_v - sum_u p_u g_u),  g = log p + 1, or log 1e-5 where the
    //                              clamp of :259 is active
    // cloning: coef = -1/N, every token valid, no entropy term -> (softmax - onehot) / N
    const bool on = lane < V;
    const bool valid = on && (a.valid_bits ? ((a.valid_bits[tn] >> lane) & 1) != 0 : true);
    const float sc = valid ? a.scores[tn * V + lane] : -INFINITY;
    const float mx = wave_max(sc);
    const float ex = valid ? expf(sc - mx) : 0.f;
    const float den = wave_sum(ex);
    const float pv = ex / den;
    const float coef = a.coef ? a.coef[n] : -a.inv_n;
    float d = valid ? coef * ((lane == a.gt[tn] ? 1.f : 0.f) - pv) : 0.f;
    if (a.ent_coef != 0.f) {
      const float gv = valid ? (pv >= 1e-5f ? logf(pv) + 1.f : logf(1e-5f)) : 0.f;
      const float pg = wave_sum(pv * gv);
      d += valid ? a.ent_coef * pv * (gv - pg) : 0.f;
    }
    if (lane < 16) {
      dscs[lane] = d;
      a.dsc[tn * 16 + lane] = d;
    }
  }
  for (int tau = tid; tau < T; tau += 256) ats[tau] = a.atts[((size_t)t * T + tau) * N + n];
  __syncthreads();
  for (int k = tid; k < 2 * L; k += 256) {
    const float* wr = a.Wy + (size_t)k * V;
    float s = 0.f;
    for (int sI = 0; sI < V; ++sI) s += wr[sI] * dscs[sI];
    if (k < L) a.dout[tn * L + k] = s;
    else { dctx[k - L] = s; a.dctx[tn * L + (k - L)] = s; }
  }
  __syncthreads();
  // datt[tau] = dctx . eout[tau, n, :] + datts_wv     (wave per tau; the rows of UT taus are
  // fetched before any is reduced -- one row per trip made every trip a full memory round trip)
  constexpr int UT = 4;
  {
    float4 d4[DB_MAXKI];
#pragma unroll
    for (int i = 0; i < DB_MAXKI; ++i) {
      const int k = 4 * lane + 256 * i;
      if (k < L) d4[i] = *reinterpret_cast<const float4*>(dctx + k);
    }
    for (int tau0 = w; tau0 < T; tau0 += 4 * UT) {
      float4 e4[UT][DB_MAXKI];
      float wvv[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int tau = tau0 + 4 * u;
        const int tr = tau < len ? tau : 0;              // clamped: loads stay unconditional
        const float* er = a.eout + ((size_t)tr * N + n) * L;
#pragma unroll
        for (int i = 0; i < DB_MAXKI; ++i) {
          const int k = 4 * lane + 256 * i;
          if (k < L) e4[u][i] = *reinterpret_cast<const float4*>(er + k);
        }
        wvv[u] = a.datts_wv[((size_t)t * T + (tau < T ? tau : 0)) * N + n];
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int tau = tau0 + 4 * u;
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < DB_MAXKI; ++i) {
          const int k = 4 * lane + 256 * i;
          if (k < L)
            sacc += e4[u][i].x * d4[i].x + e4[u][i].y * d4[i].y + e4[u][i].z * d4[i].z +
                    e4[u][i].w * d4[i].w;
        }
        sacc = wave_sum(sacc);
        if (lane == 0 && tau < T) des[tau] = (tau < len ? sacc : 0.f) + wvv[u];
      }
    }
  }
  __syncthreads();
  {
    float ls = 0.f;
    for (int tau = tid; tau < T; tau += 256) ls += ats[tau] * des[tau];
    const float sad = block_reduce<0>(ls, dscs);       // dscs no longer needed
    for (int tau = tid; tau < T; tau += 256) {
      const float d = ats[tau] * (des[tau] - sad);      // att == 0 past the length
      des[tau] = d;
      a.de[((size_t)t * T + tau) * N + n] = d;
    }
  }
  __syncthreads();
  // dq / dv partials: lanes own k (float4), waves stride over tau
  {
    float4 v4[DB_MAXKI], q4[DB_MAXKI], dq4[DB_MAXKI], dv4[DB_MAXKI];
    const float* qrow = a.q + tn * L;
#pragma unroll
    for (int i = 0; i < DB_MAXKI; ++i) {
      const int k = 4 * lane + 256 * i;
      dq4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      dv4[i] = dq4[i];
      if (k < L) {
        v4[i] = *reinterpret_cast<const float4*>(a.v + k);
        q4[i] = *reinterpret_cast<const float4*>(qrow + k);
      }
    }
    const int tend = min(len, T);
    for (int tau0 = w; tau0 < tend; tau0 += 4 * UT) {
      float4 e4[UT][DB_MAXKI];
      float dd[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) {                   // rows of UT taus in flight
        const int tau = tau0 + 4 * u;
        const int tr = tau < tend ? tau : tau0;
        dd[u] = tau < tend ? des[tau] : 0.f;           // a zero weight switches the trip off
        const float* er = a.eht + ((size_t)tr * N + n) * L;
#pragma unroll
        for (int i = 0; i < DB_MAXKI; ++i) {
          const int k = 4 * lane + 256 * i;
          if (k < L) e4[u][i] = *reinterpret_cast<const float4*>(er + k);
        }
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const float d = dd[u];
#pragma unroll
        for (int i = 0; i < DB_MAXKI; ++i) {
          const int k = 4 * lane + 256 * i;
          if (k < L) {
            const float4 e = e4[u][i];
            const float t0 = fast_tanh(q4[i].x + e.x), t1 = fast_tanh(q4[i].y + e.y),
                        t2 = fast_tanh(q4[i].z + e.z), t3 = fast_tanh(q4[i].w + e.w);
            dv4[i].x += d * t0; dv4[i].y += d * t1; dv4[i].z += d * t2; dv4[i].w += d * t3;
            dq4[i].x += d * v4[i].x * (1.f - t0 * t0); dq4[i].y += d * v4[i].y * (1.f - t1 * t1);
            dq4[i].z += d * v4[i].z * (1.f - t2 * t2); dq4[i].w += d * v4[i].w * (1.f - t3 * t3);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < DB_MAXKI; ++i) {
      const int k = 4 * lane + 256 * i;
      if (k < L) {
        *reinterpret_cast<float4*>(red + (size_t)(w * 2 + 0) * L + k) = dq4[i];
        *reinterpret_cast<float4*>(red + (size_t)(w * 2 + 1) * L + k) = dv4[i];
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < L; k += 256) {
    float sq = 0.f, sv = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { sq += red[(size_t)(ww * 2) * L + k]; sv += red[(size_t)(ww * 2 + 1) * L + k]; }
    a.dq[tn * L + k] = sq;
    a.dvp[tn * L + k] = sv;
  }
}

// dec_bwd_b: one workgroup per (encoder step tau, question n), lanes over k:
//   deht[tau,n,k]  = sum_t de[t,tau,n] v_k (1 - tanh^2(q[t,n,k] + eht[tau,n,k]))
//   deout[tau,n,k] = sum_t att[t,tau,n] dctx[t,n,k]
__global__ __launch_bounds__(128) void dec_bwd_b_kernel(DecBwdArgs a) {
  const int L = a.L, T = a.T, N = a.N;
  const int tau = blockIdx.x, n = blockIdx.y;
  const int len = min(max(a.seq_len[n], 0), T);
  const size_t row = ((size_t)tau * N + n) * L;
  for (int k = 4 * threadIdx.x; k < L; k += 512) {
    float4 dh = make_float4(0.f, 0.f, 0.f, 0.f), dob = dh;
    if (tau < len) {
      const float4 e4 = *reinterpret_cast<const float4*>(a.eht + row + k);
      const float4 v4 = *reinterpret_cast<const float4*>(a.v + k);
      for (int t = 0; t < a.Td; ++t) {
        const size_t tn = (size_t)t * N + n;
        const float d = a.de[((size_t)t * T + tau) * N + n];
        const float at = a.atts[((size_t)t * T + tau) * N + n];
        const float4 q4 = *reinterpret_cast<const float4*>(a.q + tn * L + k);
        const float4 c4 = *reinterpret_cast<const float4*>(a.dctx + tn * L + k);
        const float t0 = fast_tanh(q4.x + e4.x), t1 = fast_tanh(q4.y + e4.y),
                    t2 = fast_tanh(q4.z + e4.z), t3 = fast_tanh(q4.w + e4.w);
        dh.x += d * v4.x * (1.f - t0 * t0); dh.y += d * v4.y * (1.f - t1 * t1);
        dh.z += d * v4.z * (1.f - t2 * t2); dh.w += d * v4.w * (1.f - t3 * t3);
        dob.x += at * c4.x; dob.y += at * c4.y; dob.z += at * c4.z; dob.w += at * c4.w;
      }
    }
    *reinterpret_cast<float4*>(a.deht + row + k) = dh;
    *reinterpret_cast<float4*>(a.deout + row + k) = dob;
  }
}

// ---------------------------------------------------------------------------------------------
// word_vecs_bwd: word_vecs[t,n,:] = sum_tau atts[t,tau,n] * emb[seq[tau,n],:]
//   blockIdx.y == 0:  datts_wv[t,tau,n] = dwv[t,n,:] . emb[seq[tau,n],:]          (tau < len, else 0)
//   blockIdx.y == 1:  dE[tau,n,:] = sum_t atts[t,tau,n] * dwv[t,n,:]              (tau < len only)
// dE is the gradient of the embedded question; the embedding-matrix gradient is then the one-hot
// gemm_tn over the active rows (no atomics).  One workgroup per (question, half); the question's
// embedding rows (row stride E+1: conflict-free across tau) and dwv rows are staged in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void word_vecs_bwd_kernel(
    const float* __restrict__ dwv, const float* __restrict__ atts, const int32_t* __restrict__ seq,
    const int32_t* __restrict__ seq_len, const float* __restrict__ emb, int T_dec, int T_enc,
    int N, int E, float* __restrict__ datts_wv, float* __restrict__ dE) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ES = E + 1;
  float* dw = smem;                            // [T_dec][E]
  float* rows = dw + (size_t)T_dec * E;        // y == 0: [T_enc][E+1]   y == 1: at [T_dec][T_enc]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int len = min(seq_len[n], T_enc);
  for (int i = tid; i < T_dec * E; i += 256) {
    const int t = i / E, e = i - t * E;
    dw[i] = dwv[((size_t)t * N + n) * E + e];
  }
  if (blockIdx.y == 0) {
    for (int i = tid; i < len * E; i += 256) {
      const int tau = i / E, e = i - tau * E;
      rows[tau * ES + e] = emb[(size_t)seq[tau * N + n] * E + e];
    }
    __syncthreads();
    for (int p = tid; p < T_dec * T_enc; p += 256) {          // thread per (t, tau)
      const int t = p / T_enc, tau = p - t * T_enc;
      float s0 = 0.f, s1 = 0.f;
      if (tau < len) {
        const float* r = rows + tau * ES;
        const float* d = dw + t * E;
        int e = 0;
        for (; e + 1 < E; e += 2) { s0 += r[e] * d[e]; s1 += r[e + 1] * d[e + 1]; }
        if (e < E) s0 += r[e] * d[e];
      }
      datts_wv[((size_t)t * T_enc + tau) * N + n] = s0 + s1;
    }
  } else {
    float* at = rows;
    for (int i = tid; i < T_dec * T_enc; i += 256) {
      const int t = i / T_enc, tau = i - t * T_enc;
      at[i] = atts[((size_t)t * T_enc + tau) * N + n];
    }
    __syncthreads();
    for (int i = tid; i < len * E; i += 256) {
      const int tau = i / E, e = i - tau * E;
      float s = 0.f;
      for (int t = 0; t < T_dec; ++t) s += at[t * T_enc + tau] * dw[t * E + e];
      dE[((size_t)tau * N + n) * E + e] = s;
    }
  }
}

// losses[0] = mean_n CE(scores[n], label[n]);  losses[1] = mean_n(-log_seq_prob[n])
// dscores[n][c] = (softmax(scores[n])[c] - [c == label]) / N
// (tf.nn.sparse_softmax_cross_entropy_with_logits + reduce_mean, train_clevr_gt_layout.py:104-111)
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ scores,
                                                   const int32_t* __restrict__ labels,
                                                   const float* __restrict__ log_seq_prob, int N,
                                                   int C, float* __restrict__ dscores,
                                                   float* __restrict__ losses,
                                                   float* __restrict__ ds_pad, int Cp) {
  // one wave per question (4 per workgroup, one workgroup): lanes stride the classes
  __shared__ float part[4][2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float ce = 0.f, nl = 0.f;
  for (int n = w; n < N; n += 4) {
    const float* z = scores + (size_t)n * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, z[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(z[c] - mx);
    s = wave_sum(s);
    const int lab = labels[n];
    for (int c = lane; c < C; c += 64) {
      const float d = (expf(z[c] - mx) / s - (c == lab ? 1.f : 0.f)) / (float)N;
      dscores[(size_t)n * C + c] = d;
      if (ds_pad) ds_pad[(size_t)n * Cp + c] = d;
    }
    if (ds_pad) for (int c = C + lane; c < Cp; c += 64) ds_pad[(size_t)n * Cp + c] = 0.f;
    if (lane == 0) { ce += logf(s) + mx - z[lab]; nl -= log_seq_prob[n]; }
  }
  if (lane == 0) { part[w][0] = ce; part[w][1] = nl; }
  __syncthreads();
  if (threadIdx.x == 0) {
    losses[0] = (part[0][0] + part[1][0] + part[2][0] + part[3][0]) / (float)N;
    losses[1] = (part[0][1] + part[1][1] + part[2][1] + part[3][1]) / (float)N;
  }
}

// Dropout multipliers from a counter-based generator: element i of the stream (seed, offset + i)
// is splitmix64 of its counter, so any slice can be regenerated independently (no state).
__global__ void dropout_mult_kernel(float* __restrict__ out, size_t n, float keep_prob,
                                    unsigned long long seed, unsigned long long offset) {
  const float scale = 1.0f / keep_prob;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = seed + (offset + i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // 24 bits -> [0, 1)
    out[i] = u < keep_prob ? scale : 0.f;
  }
}

// dst[idx[rows[i]]][:] += src[rows[i]][:] for i < *count: the embedding-matrix gradient from the
// per-position embedding gradients (large vocabularies; the one-hot GEMM of the small-vocabulary
// path would contract over the whole vocabulary).  One wave per row.
__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ src,
                                                            const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ rows,
                                                            const int32_t* __restrict__ count,
                                                            int ncols, float* __restrict__ dst) {
  const int n = *count;
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
    const int r = rows[i];
    const float* s = src + (size_t)r * ncols;
    float* d = dst + (size_t)idx[r] * ncols;
    for (int c = lane; c < ncols; c += 64) atomicAdd(d + c, s[c]);
  }
}

// small elementwise helpers of the question-prior-net step (models_vqa/question_prior_net.py:22-27)
__global__ void ew_mul_kernel(float* __restrict__ x, const float* __restrict__ m, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] *= m[i];
}
// d fc1 pre-activation = d(dropped relu output) * dropout multiplier * [relu output > 0]
__global__ void qpn_dpre_kernel(float* __restrict__ dad, const float* __restrict__ ad,
                                const float* __restrict__ m1, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dad[i] = ad[i] > 0.f ? dad[i] * (m1 ? m1[i] : 1.f) : 0.f;
}
// dH0[n][u] += dh[n][u] * mh, dH1[n][u] += dh[n][L + u] * mh: the question prior reads the final
// hidden states of both encoder layers
__global__ void qpn_dh_add_kernel(const float* __restrict__ dh, const float* __restrict__ mh,
                                  float* __restrict__ dH0, float* __restrict__ dH1, int N, int L) {
  const size_t n2 = (size_t)N * 2 * L;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2;
       i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / (2 * L)), j = (int)(i - (size_t)n * 2 * L);
    const float v = dh[i] * (mh ? mh[i] : 1.f);
    if (j < L) dH0[(size_t)n * L + j] += v; else dH1[(size_t)n * L + j - L] += v;
  }
}

// policy-gradient objective (exp_clevr/train_clevr_rl_gt_layout.py:107-129):
//   final[n] = validity[n] ? CE(scores[n], label[n]) : invalid_expr_loss
//   losses[0] = mean(final); losses[1] = mean((final - baseline) * log_seq_prob);
//   losses[4] = mean(neg_entropy);  coef[n] = (final[n] - baseline) / N  (d total / d log_seq_prob)
//   dscores[n] = validity[n] ? (softmax - onehot) / N : 0;  then baseline += (1-decay)(mean - baseline)
__global__ __launch_bounds__(256) void loss_rl_kernel(LossRlArgs a) {
  __shared__ float scratch[16];
  const int N = a.N, C = a.C;
  const float base = *a.baseline;
  float fl = 0.f, pg = 0.f, en = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float* z = a.scores + (size_t)n * C;
    const bool ok = a.expr_validity[n] != 0;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - mx);
    const int lab = a.labels[n];
    const float fin = ok ? logf(s) + mx - z[lab] : a.invalid_expr_loss;
    for (int c = 0; c < C; ++c)
      a.dscores[(size_t)n * C + c] =
          ok ? (expf(z[c] - mx) / s - (c == lab ? 1.f : 0.f)) / (float)N : 0.f;
    a.coef[n] = (fin - base) / (float)N;
    fl += fin;
    pg += (fin - base) * a.log_seq_prob[n];
    en += a.neg_entropy[n];
  }
  const float tfl = block_reduce<0>(fl, scratch);
  const float tpg = block_reduce<0>(pg, scratch);
  const float ten = block_reduce<0>(en, scratch);
  if (threadIdx.x == 0) {
    const float avg = tfl / (float)N;
    a.losses[0] = avg;
    a.losses[1] = tpg / (float)N;
    a.losses[4] = ten / (float)N;
    *a.baseline = base + (1.f - a.baseline_decay) * (avg - base);
  }
}

// total_loss = seq_likelihood_loss | policy_gradient_loss + avg_sample_loss
//              + lambda_entropy * entropy_reg + weight_decay * l2_reg
// (train_clevr_gt_layout.py:113-114, train_clevr_rl_gt_layout.py:126-129)
__global__ void loss_total_kernel(float* __restrict__ losses, float wd, float lambda_entropy) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    losses[3] = losses[0] + losses[1] + lambda_entropy * losses[4] + wd * losses[2];
}

// ---------------------------------------------------------------------------------------------
// optimiser (train_clevr_gt_layout.py:112-120): total_loss includes weight_decay * l2_reg over the
// variables named '.../weights' (nmn3_model.py:161-166); per-tensor tf.clip_by_norm(g, 10);
// tf.train.AdamOptimizer() with its TF 1.0.0 defaults.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_finish_kernel(float* __restrict__ grads,
                                                          const float* const* __restrict__ mirrors,
                                                          const int64_t* __restrict__ var_off,
                                                          const int32_t* __restrict__ decay,
                                                          const ParamSeg* __restrict__ segs,
                                                          float scale, float wd,
                                                          float* __restrict__ l2_out) {
  __shared__ float scratch[16];
  const ParamSeg sg = segs[blockIdx.x];
  const bool dec = decay[sg.var] != 0;
  const float* wsrc = mirrors[sg.var] - var_off[sg.var];
  float l2 = 0.f;
  constexpr int U = 8;      // independent loads per thread in flight (a plain strided loop ran at
                            // 1.7 TB/s: one 4-B load per trip)
  for (int64_t base = sg.begin + threadIdx.x; base < sg.end; base += 256 * U) {
    float gv[U], wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + 256 * u;
      const int64_t ii = i < sg.end ? i : sg.begin;
      gv[u] = grads[ii];
      wv[u] = dec ? wsrc[ii] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + 256 * u;
      if (i < sg.end) {
        grads[i] = gv[u] * scale + wd * wv[u];
        l2 += 0.5f * wv[u] * wv[u];
      }
    }
  }
  const float t = block_reduce<0>(l2, scratch);
  if (dec && threadIdx.x == 0) atomicAdd(l2_out, t);
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ grads,
                                                          const ParamSeg* __restrict__ segs,
                                                          float scale, float* __restrict__ norm2) {
  __shared__ float scratch[16];
  const ParamSeg sg = segs[blockIdx.x];
  float s = 0.f;
  constexpr int U = 8;
  for (int64_t base = sg.begin + threadIdx.x; base < sg.end; base += 256 * U) {
    float gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + 256 * u;
      gv[u] = i < sg.end ? grads[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { const float g = gv[u] * scale; s += g * g; }
  }
  const float t = block_reduce<0>(s, scratch);
  if (threadIdx.x == 0) atomicAdd(norm2 + sg.var, t);
}

__global__ __launch_bounds__(256) void adam_kernel(const float* __restrict__ grads,
                                                   float* const* __restrict__ mirrors,
                                                   const int64_t* __restrict__ var_off,
                                                   const ParamSeg* __restrict__ segs,
                                                   const float* __restrict__ norm2, float scale,
                                                   float clip, float lr_t, float beta1, float beta2,
                                                   float eps, float* __restrict__ m,
                                                   float* __restrict__ v) {
  const ParamSeg sg = segs[blockIdx.x];
  // tf.clip_by_norm: g * clip / max(||g||, clip)
  const float cs = scale * clip / fmaxf(sqrtf(norm2[sg.var]), clip);
  float* wdst = mirrors[sg.var] - var_off[sg.var];
  // blockIdx.y: quarter of the segment (the segments are sized for the read-only passes; seven
  // streams per element want more workgroups in flight)
  const int64_t qlen = (sg.end - sg.begin + gridDim.y - 1) / gridDim.y;
  const int64_t qbeg = sg.begin + blockIdx.y * qlen, qend = min(sg.end, qbeg + qlen);
  constexpr int U = 4;
  for (int64_t base = qbeg + threadIdx.x; base < qend; base += 256 * U) {
    float gv[U], mv[U], vv[U], wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + 256 * u;
      const int64_t ii = i < qend ? i : sg.begin;
      gv[u] = grads[ii]; mv[u] = m[ii]; vv[u] = v[ii]; wv[u] = wdst[ii];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + 256 * u;
      if (i < qend) {
        const float g = gv[u] * cs;
        const float mi = beta1 * mv[u] + (1.f - beta1) * g;
        const float vi = beta2 * vv[u] + (1.f - beta2) * g * g;
        m[i] = mi;
        v[i] = vi;
        wdst[i] = wv[u] - lr_t * mi / (sqrtf(vi) + eps);
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_gemm_tn(const GemmTnArgs& a_in, hipStream_t s, int max_resident) {
  if (a_in.M <= 0 || a_in.N <= 0 || a_in.R <= 0) return;
  GemmTnArgs a = a_in;
  const int nprob = a.nprob > 0 ? a.nprob : 1;
  constexpr int TB = 64, BK = 32;
  const int gx = (a.N + TB - 1) / TB, gy = (a.M + TB - 1) / TB;
  // Four workgroups fit a CU (1024 on the chip).  Split the reduction while the launch stays
  // within that and a split keeps >= 4 k-tiles: every split costs M*N atomic adds, and a single
  // split writes its tile with plain read-modify-writes.  (Whole training step, same box: 2.917 ms
  // with this rule, 2.954 with "split until >= 1024 workgroups", 2.969 with a 2048 target.)
  int splits = 1;
  const int nkt = (a.R + BK - 1) / BK;
  while (gx * gy * nprob * splits * 2 <= 1024 && nkt / (splits * 2) >= 4) splits *= 2;
  int r_per = ((nkt + splits - 1) / splits) * BK;
  splits = (a.R + r_per - 1) / r_per;
  const int Z = splits * nprob;
  a.z_total = Z;
  int z_per = Z;
  if (max_resident > 0 && gx * gy * Z > max_resident) z_per = std::max(1, max_resident / (gx * gy));
  for (int z0 = 0; z0 < Z; z0 += z_per) {
    a.z_off = z0;
    hipLaunchKernelGGL((gemm_tn_kernel<1, 32>), dim3(gx, gy, std::min(z_per, Z - z0)), dim3(256), 0, s,
                       a, r_per);
  }
}

void launch_zero_ranges(const ZeroRanges& z, hipStream_t s) {
  if (z.n <= 0) return;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(2048), dim3(256), 0, s, z);
}

void launch_active_rows(const int32_t* seq_len, int T, int N, int32_t* rows, int32_t* count,
                        int32_t* rows_ch, const int* chunk_start, hipStream_t s) {
  const int4 st = chunk_start ? make_int4(chunk_start[0], chunk_start[1], chunk_start[2], chunk_start[3])
                              : make_int4(0, 0, 0, 0);
  hipLaunchKernelGGL(active_rows_kernel, dim3((T * N + 255) / 256), dim3(256), 0, s, seq_len, T, N,
                     rows, count, chunk_start ? rows_ch : nullptr, st);
}

void launch_colsum(const float* src, int R, int ncols, int ld, const int32_t* sel, int sel_val,
                   float* dst, hipStream_t s) {
  if (R <= 0 || ncols <= 0) return;
  const int gx = (ncols + 63) / 64;
  int gy = 1;
  while (gx * gy < 256 && R / (gy * 2) >= 32) gy *= 2;
  const int r_per = (R + gy - 1) / gy;
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, s, src, R, ncols, ld, sel, sel_val,
                     dst, r_per);
}

void launch_pack_pk_t(const float* src, int ld, int K, int N, float* dst, int Kp, int Np,
                      hipStream_t s) {
  const size_t total = (size_t)Kp * Np;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_pk_t_kernel, dim3(blocks), dim3(256), 0, s, src, ld, K, N, dst, Kp, Np);
}

void launch_pack_tiles_t(const float* W, int ld, int row0, int L, float* dst, int Ktot, int k_off,
                         hipStream_t s) {
  const size_t total = (size_t)(L / 16) * L * 64;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_tiles_t_kernel, dim3(blocks), dim3(256), 0, s, W, ld, row0, L, dst, Ktot,
                     k_off);
}

void launch_lstm_bwd_step(const LstmBwdJob* jobs, int njobs, int N, int L, hipStream_t s) {
  LstmBwdJobs js;
  for (int i = 0; i < 2; ++i) {
    if (i < njobs) js.j[i] = jobs[i];
    else { js.j[i] = LstmBwdJob{}; js.j[i].active = 0; }
  }
  // enough workgroups from the column tiles alone (lstm_dim >= 1024) -> 32-row tiles
  if (L / 16 * njobs >= 128 && N > 16) {
    dim3 grid(L / 16, njobs, (N + 31) / 32);
    hipLaunchKernelGGL((lstm_bwd_step_kernel<2, 8>), grid, dim3(512), 0, s, js, N, L);
  } else {
#ifndef BWS_WAVES
#define BWS_WAVES 8
#endif
    static const int waves = N2NMN_KNOB_INT("N2NMN_BWD_WAVES", BWS_WAVES) == 16 ? 16 : 8;
    dim3 grid(L / 16, njobs, (N + 15) / 16);
    if (waves == 16)
      hipLaunchKernelGGL((lstm_bwd_step_kernel<1, 16>), grid, dim3(1024), 0, s, js, N, L);
    else
      hipLaunchKernelGGL((lstm_bwd_step_kernel<1, 8>), grid, dim3(512), 0, s, js, N, L);
  }
}

void launch_dec_xidx(const int32_t* gt, int Td, int N, int go_row, int32_t* idx, hipStream_t s) {
  hipLaunchKernelGGL(dec_xidx_kernel, dim3((Td * N + 255) / 256), dim3(256), 0, s, gt, Td, N,
                     go_row, idx);
}

void launch_dec_bwd_a(const DecBwdArgs& a, hipStream_t s) {
  const size_t smem = sizeof(float) * ((size_t)a.L + 2 * ((a.T + 3) & ~3) + 16 + 8 * (size_t)a.L);
  hipLaunchKernelGGL(dec_bwd_a_kernel, dim3(a.N, a.Td), dim3(256), smem, s, a);
}

void launch_dec_bwd_b(const DecBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(dec_bwd_b_kernel, dim3(a.T, a.N), dim3(128), 0, s, a);
}

void launch_word_vecs_bwd(const float* dwv, const float* atts, const int32_t* seq,
                          const int32_t* seq_len, const float* emb, int T_dec, int T_enc, int N,
                          int E, float* datts_wv, float* dE, hipStream_t s) {
  const size_t smem = sizeof(float) * ((size_t)T_dec * E +
                                       std::max((size_t)T_enc * (E + 1), (size_t)T_dec * T_enc) + 4);
  if (smem > 64 * 1024)      // a workgroup may use the whole 160 KiB LDS of a gfx950 CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(word_vecs_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(word_vecs_bwd_kernel, dim3(N, 2), dim3(256), smem, s, dwv, atts, seq, seq_len,
                     emb, T_dec, T_enc, N, E, datts_wv, dE);
}

void launch_loss(const float* scores, const int32_t* labels, const float* log_seq_prob, int N,
                 int C, float* dscores, float* losses, hipStream_t s, float* ds_pad, int Cp) {
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, s, scores, labels, log_seq_prob, N, C,
                     dscores, losses, ds_pad, Cp);
}

void launch_embed_scatter(const float* src, const int32_t* idx, const int32_t* rows,
                          const int32_t* count, int max_rows, int ncols, float* dst, hipStream_t s) {
  if (max_rows <= 0) return;
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(std::min((max_rows + 3) / 4, 1024)), dim3(256), 0, s,
                     src, idx, rows, count, ncols, dst);
}

static int ew_blocks(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 1024); }
void launch_dropout_mult(float* out, size_t n, float keep_prob, unsigned long long seed,
                         unsigned long long offset, hipStream_t s) {
  hipLaunchKernelGGL(dropout_mult_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, out, n, keep_prob, seed,
                     offset);
}
void launch_ew_mul(float* x, const float* m, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(ew_mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, m, n);
}
void launch_qpn_dpre(float* dad, const float* ad, const float* m1, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(qpn_dpre_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dad, ad, m1, n);
}
void launch_qpn_dh_add(const float* dh, const float* mh, float* dH0, float* dH1, int N, int L,
                       hipStream_t s) {
  hipLaunchKernelGGL(qpn_dh_add_kernel, dim3(ew_blocks((size_t)N * 2 * L)), dim3(256), 0, s, dh, mh,
                     dH0, dH1, N, L);
}

void launch_loss_rl(const LossRlArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(loss_rl_kernel, dim3(1), dim3(256), 0, s, a);
}

void launch_loss_total(float* losses, float wd, float lambda_entropy, hipStream_t s) {
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, s, losses, wd, lambda_entropy);
}

void launch_grad_finish(float* grads, const float* const* mirrors, const int64_t* var_off,
                        const int32_t* decay, const ParamSeg* segs, int nsegs, float scale,
                        float wd, float* l2_out, hipStream_t s) {
  if (nsegs <= 0) return;
  hipLaunchKernelGGL(grad_finish_kernel, dim3(nsegs), dim3(256), 0, s, grads, mirrors, var_off,
                     decay, segs, scale, wd, l2_out);
}

void launch_grad_sqnorm(const float* grads, const ParamSeg* segs, int nsegs, float scale,
                        float* norm2, hipStream_t s) {
  if (nsegs <= 0) return;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nsegs), dim3(256), 0, s, grads, segs, scale, norm2);
}

void launch_adam(const float* grads, float* const* mirrors, const int64_t* var_off,
                 const ParamSeg* segs, int nsegs, const float* norm2, float scale, float clip,
                 float lr_t, float beta1, float beta2, float eps, float* m, float* v,
                 hipStream_t s) {
  if (nsegs <= 0) return;
  hipLaunchKernelGGL(adam_kernel, dim3(nsegs, 4), dim3(256), 0, s, grads, mirrors, var_off, segs,
                     norm2, scale, clip, lr_t, beta1, beta2, eps, m, v);
}

}  // namespace n2nmn
