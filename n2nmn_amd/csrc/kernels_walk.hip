// Layout walker: the whole module network of ONE question in ONE workgroup, straight from the
// decoder's Reverse-Polish tokens -- on-device replacement of
//   Assembler.assemble               models_clevr/nmn3_assembler.py:153-222  (stack decode + validity)
//   td.Compiler.build_feed_dict/Loom models_clevr/nmn3_model.py:55-159       (per-depth batching)
//   Modules.<X>Module                models_clevr/nmn3_modules.py:60-495
// for the inference path.
//
// Why a walker instead of per-(level, stage) launches: questions are independent and a CLEVR layout
// has <= T_dec nodes whose attention maps are 150 floats each, so a question's whole tree fits one
// workgroup's LDS.  The level scheduler (schedule.cpp) pays a kernel boundary per level stage and its
// launches carry ~15 jobs each; here the grid is (questions of ALL in-flight batches) workgroups of
// 512 threads, every operator runs as soon as its inputs exist, nothing but the answer logits is
// written back, and no host synchronisation or program upload sits between the two phases.
//
// Per workgroup (question n of batch k):
//   1. thread 0 decodes the RPN column tokens[:, n] with the reference's five validity checks;
//      an invalid layout writes zero logits (nmn3_model.py:146,155) and validity = 0.
//   2. nodes run in token order (= a valid topological order); node t keeps its attention map in
//      LDS slot t.  Text maps (fc_text) are computed on the fly from word_vecs[t, n].
//   3. Find / Filter / FindSameProperty read the hoisted conv_image map of image n (one MFMA GEMM
//      per batch, run beside phase 1), the pooling operators stream feat[n] (the HBM-bound read:
//      every thread has all of its 16-B loads in flight before the softmax), answer operators write
//      scores[n, :].
//
// HBM traffic per question: conv_image map 150 KB per Find-type node + 307 KB per pooling node
// (+ weights from L2); algorithmic bytes are counted by n2nmn_walk_layouts for the roofline.
#include <algorithm>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

constexpr int WT = WALK_THREADS;       // 512: 8 waves with up to 256 VGPRs each (one workgroup per CU)
constexpr int WW = WT / 64;            // 8 waves
constexpr int MAXT = WALK_MAX_T;       // decoder steps a layout may have
constexpr int MAXCI = 4;               // float4 column groups per lane: Mp <= 256 * CI, CI <= 4

struct WalkLds {
  int op[MAXT], in0[MAXT], in1[MAXT];
  int tok_op[MAXT];        // op code of token t (-1: <eos>, -2: token out of range)
  int stack[MAXT];
  int flist[MAXT];         // nodes that read the FindModule conv_image map (Find, Filter)
  int n_nodes, valid, n_find;
};


// block reductions through `scr` (>= 16 floats); every thread gets the result
template <int OP>
__device__ __forceinline__ float wg_reduce(float v, float* scr) {
  return block_reduce<OP>(v, scr);
}

// out[c] (c < Mp, LDS) = bias[c] + sum_k x[k] * Wp[k][c]   with Wp zero-padded [K][Mp] in HBM/L2,
// x in LDS.  K-split over the waves, float4 columns over the lanes; every lane has all of its
// weight rows in flight before the first FMA (<= KU rows per pass).  `red` needs WW*Mp floats.
// scale (LDS, may be nullptr): out[c] *= scale[c] after the bias.
template <int KU>
__device__ __forceinline__ void fc_pad(int tid, const float* x, int K, const float* __restrict__ Wp,
                                       const float* __restrict__ bias, int Mp, float* out,
                                       float* red, const float* scale) {
  const int lane = tid & 63, wid = tid >> 6;
  const int kper = (K + WW - 1) / WW;
  const int k0 = wid * kper, k1 = min(K, k0 + kper);
  for (int cb = 0; cb < Mp; cb += 256) {
    const bool col_ok = cb + 4 * lane < Mp;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // uniform base (SGPR pair) + 32-bit per-lane offset: one address VGPR per load in flight
    const unsigned col = (unsigned)min(cb + 4 * lane, Mp - 4);
    for (int kb = k0; kb < k1; kb += KU) {
      float4 w4[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const unsigned k = (unsigned)min(kb + u, k1 - 1);
        w4[u] = *reinterpret_cast<const float4*>(Wp + (k * (unsigned)Mp + col));
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (kb + u < k1) {
          const float xv = x[kb + u];
          acc.x += xv * w4[u].x; acc.y += xv * w4[u].y; acc.z += xv * w4[u].z; acc.w += xv * w4[u].w;
        }
      }
    }
    if (col_ok) *reinterpret_cast<float4*>(red + (size_t)wid * 256 + 4 * lane) = acc;
    __syncthreads();
    if (tid < 256 && cb + tid < Mp) {
      float r = bias[cb + tid];
#pragma unroll
      for (int q = 0; q < WW; ++q) r += red[q * 256 + tid];
      if (scale) r *= scale[cb + tid];
      out[cb + tid] = r;
    }
    __syncthreads();
  }
}

// scores[c] = b[c] + sum_f x[f] * Wm[f*C + c], x in LDS (F values), C <= WT.  Thread (slice, c)
// takes f = slice, slice + nsl, ...: consecutive threads read consecutive floats of Wm.
// The thread's weights go out in groups of FC_FU loads before the first multiply (a loop that loads and
// accumulates pays an L2 round trip per trip: 9 for CountModule's 152 features, a third of a question's
// whole chain).  Even trips accumulate into s0, odd trips into s1.  The FIRST group can be requested by
// the caller long before x exists (fc_out_first: walk_light_kernel asks for it as soon as the plan names
// the root operator, the round trip then runs under the tree evaluation) -- same sums in the same order.
constexpr int FC_FU = 10;
struct FcFirst { float v[FC_FU]; };
template <int NT = WT>
__device__ __forceinline__ FcFirst fc_out_first(int tid, int F, const float* __restrict__ Wm, int C) {
  const int nsl = NT / C;
  const int c = tid % C, sl = tid / C;
  FcFirst r;
#pragma unroll
  for (int u = 0; u < FC_FU; ++u)
    r.v[u] = sl < nsl ? Wm[(size_t)min(sl + u * nsl, F - 1) * C + c] : 0.f;
  return r;
}
template <bool PRE, int NT = WT>
__device__ __forceinline__ void fc_out_t(int tid, const float* x, int F, const float* __restrict__ Wm,
                                         const float* __restrict__ b, int C, float* __restrict__ out,
                                         float* red, const FcFirst& pre) {
  const int nsl = NT / C;
  const int c = tid % C, sl = tid / C;
  float s0 = 0.f, s1 = 0.f;
  if (sl < nsl) {
    constexpr int FU = FC_FU;
    for (int f0 = sl; f0 < F; f0 += FU * nsl) {
      float wv[FU];
      if (PRE && f0 == sl) {
#pragma unroll
        for (int u = 0; u < FU; ++u) wv[u] = pre.v[u];
      } else {
#pragma unroll
        for (int u = 0; u < FU; ++u) wv[u] = Wm[(size_t)min(f0 + u * nsl, F - 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int f = f0 + u * nsl;
        if (f < F) {
          if (u & 1) s1 += x[f] * wv[u]; else s0 += x[f] * wv[u];
        }
      }
    }
    red[sl * C + c] = s0 + s1;
  }
  __syncthreads();
  if (tid < C) {
    float r = b[tid];
    for (int q = 0; q < nsl; ++q) r += red[q * C + tid];
    out[tid] = r;
  }
  __syncthreads();
}
__device__ __forceinline__ void fc_out(int tid, const float* x, int F, const float* __restrict__ Wm,
                                       const float* __restrict__ b, int C, float* __restrict__ out,
                                       float* red) {
  FcFirst none{};
  fc_out_t<false>(tid, x, F, Wm, b, C, out, red, none);
}

}  // namespace

// LDS carve (floats) shared by host and device
__host__ __device__ inline size_t walk_lds_floats(int T, int HWp, int Mp, int E, int D, int M,
                                                  int ksize, int H, int W, int C, int T_enc = 0) {
  const int HW = H * W;
  const size_t arena = (size_t)T * HWp;
  const size_t vecs = 4 * (size_t)Mp + ((E + 3) & ~3) + 2 * (size_t)D + 2 * (size_t)HWp + 32 +
                      (size_t)((T_enc + 3) & ~3) + (T_enc > 0 ? (size_t)T * (Mp + ((T_enc + 3) & ~3)) : 0);
  // scratch: max of  fc reductions [WW][256] | pool stage [rows in flight][2][D] |
  //                  Transform taps [M][RS] + padded map + red [WW][64][2] | answer features + red
  const int KK = ksize * ksize, RS = (KK + 2 + 3) & ~3, pad = ksize / 2;
  const int ncol = D / 4, nrow = WT / ncol;
  size_t s0 = (size_t)WW * 256;
  size_t s1 = (size_t)nrow * 2 * D;
  const int KD = (KK + 1 + 3) & ~3, Mq = (M + 15) / 16 * 16, Pq = (HW + 15) / 16 * 16;
  size_t s2 = (size_t)Mq + (((size_t)(H + 2 * pad) * (W + 2 * pad) + 4) & ~3) + (size_t)WW * Pq * 2;
  (void)RS; (void)KD;
  size_t s3 = (size_t)((2 * HW + 4 + 3) & ~3) + (size_t)WT;
  size_t s = s0 > s1 ? s0 : s1;
  s = s > s2 ? s : s2;
  s = s > s3 ? s : s3;
  return arena + vecs + s + 64;
}

namespace {

// Transform (nmn3_modules.py:185-216): KSxKS SAME convolution of the 1-channel attention map to M
// channels, times the text map, l2-normalise over channels, dot with w_e -- as ONE small GEMM on the
// matrix cores.  With u[c, p] = sum_k A[c, k] * X[k, p], A[c, :] = [taps(c), bias(c), 0..] (weights
// only: packed k-major at commit time, ModuleWeights::trA) and X[:, p] = [KSxKS window of pixel p, 1,
// 0..], the text map enters only the epilogue:
//     att[p] = (sum_c tm[c] w_e[c] u[c,p]) / sqrt(max(sum_c (tm[c] u[c,p])^2, eps)) + b_e.
// Wave (g = w & 3, h = w >> 2) owns the channel tiles g, g + 4, ... and HALF of the pixel tiles (PTW of
// them): its A fragments come straight from L2 (64-B segments), the B fragments straight from the
// zero-padded map in LDS (no im2col buffer: per-lane window offsets), its PTW pixel tiles accumulate in
// independent MFMA chains, and u is folded into per-pixel partial sums in the MFMA's own output layout
// (a lane holds 4 channels of one pixel), so it never leaves registers.
// Round 5: a wave used to hold ALL pixel tiles of two channel tiles (70 B-fragment registers + 40
// accumulators: > 128 VGPRs, one workgroup per CU, the matrix phase at twice its MFMA issue time because
// the 8 waves of the only resident workgroup walk through operand build / MFMAs / folds in lock step).
// With half the pixel tiles per wave the function fits 128 VGPRs, walk_heavy_kernel runs TWO workgroups
// per CU, and one job's operand build and fold phases hide under the other's MFMAs.
// History: VALU, taps broadcast from LDS: 45 us per node; 6 interleaved FMA chains: 13 us;
// MFMA with per-node operand build + one chain: 15 us; 10 chains per wave, one workgroup per CU: 11.6 us.
// NW = 8: one workgroup owns the node (walk_kernel: wave = (channel group, pixel half)).  NW = 4: the
// workgroup owns pixel half `half` of the node and its four waves are the channel groups
// (walk_heavy_kernel: a node is two work items).  tw_ready: tm (.) w_e already in LDS (or nullptr).
constexpr int TR_CG = 4;                         // channel groups
template <int KS, int PTW, int NW>
__device__ __forceinline__ void walk_transform_t(int tid, const ModuleWeights& w, const WalkArgs& a,
                                                 const float* in0, const float* tm, const float* tw_ready,
                                                 float* outp, float* scr, long long* tl, int half) {
  // scr: [Mq] tm (.) w_e (unless tw_ready) | zero-padded map | fold buffer
  constexpr int NT = NW * 64;
  constexpr int KK = KS * KS;
  constexpr int KD = (KK + 1 + 3) & ~3;          // taps + bias row, padded to the MFMA's k = 4
  constexpr int NS = KD / 4;                     // k-steps
  constexpr int PAD = KS / 2;
  const int H = a.H, W = a.W, HW = H * W, M = a.M;
  const int PW = W + 2 * PAD, PH = H + 2 * PAD;
  const int Mt = (M + 15) >> 4, Pt = (HW + 15) >> 4;     // 16-wide channel / pixel tiles
  const int Mq = Mt * 16;
  constexpr int Pq = (NW / TR_CG) * PTW * 16;    // pixels the workgroup covers
  float* twl = scr;                              // [Mq] tm[c] * w_e[c] (both zero padded to Mp >= Mq)
  float* xin = tw_ready ? scr : twl + Mq;        // [PH][PW] zero-padded input map (+ 1 spare = 0)
  float* red = xin + ((PH * PW + 4) & ~3);       // [TR_CG][Pq][2]
  const int lane = tid & 63, wid = tid >> 6;
  const int ci = lane & 15, kg = lane >> 4;
  const int cw = __builtin_amdgcn_readfirstlane(wid);
  const int g = cw & (TR_CG - 1), pt0 = (NW == 8 ? cw / TR_CG : half) * PTW;
  const int p_base = NW == 8 ? 0 : half * PTW * 16;
  // A fragments of this wave's first channel tile: issued before anything else
  float af[NS];
  {
    const int ct0 = min(g, Mt - 1);
#pragma unroll
    for (int sI = 0; sI < NS; ++sI) af[sI] = w.trA[(4 * sI + kg) * Mq + 16 * ct0 + ci];
  }
  // tm (.) w_e for the fold, once per node (it used to ride in registers per channel tile, prefetched
  // like the A fragments: 8 VGPRs of a 128-VGPR budget)
  if (tw_ready) twl = const_cast<float*>(tw_ready);
  else for (int c = tid; c < Mq; c += NT) twl[c] = tm[c] * w.we[2][c];
  for (int i = tid; i <= PH * PW; i += NT) {
    const int y = i / PW - PAD, x = i % PW - PAD;
    xin[i] = (i < PH * PW && y >= 0 && y < H && x >= 0 && x < W) ? in0[y * W + x] : 0.f;
  }
  __syncthreads();
  if (tl && threadIdx.x == 0) tl[1] = clock64();       // debug timeline: operands ready
  // B fragments are the same for every channel tile: read them from LDS once.  Window offsets per lane:
  // k = 4*sI + kg -> (dy, dx); the bias row is a constant 1, rows beyond it 0
  float xf[NS][PTW];
  {
    const float invW = 1.0f / (float)W;
    int poff[PTW];
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) {
      const int p = min(16 * (pt0 + pt) + ci, HW - 1);
      int y = (int)(((float)p + 0.5f) * invW);       // p / W without the 40-instruction integer division
      y -= y * W > p;                                // (exact for these small values; the fix-ups make it so)
      y += (y + 1) * W <= p;
      poff[pt] = y * PW + (p - y * W);
    }
#pragma unroll
    for (int sI = 0; sI < NS; ++sI) {
      const int k = 4 * sI + kg;
      const int dy = k / KS, dx = k - dy * KS;
      const int koff = k < KK ? dy * PW + dx : 0;
      const float kmask = k < KK ? 1.f : 0.f, kone = k == KK ? 1.f : 0.f;
#pragma unroll
      for (int pt = 0; pt < PTW; ++pt) xf[sI][pt] = xin[poff[pt] + koff] * kmask + kone;
    }
  }
  float ssp[PTW], dtp[PTW];
#pragma unroll
  for (int pt = 0; pt < PTW; ++pt) { ssp[pt] = 0.f; dtp[pt] = 0.f; }
  // (debug timeline, second row of the node = row t + 16: [0] B fragments built, [1] / [2] first / last
  // channel tile of wave 0 done, [3] folds done)
  long long* tl2 = tl ? tl + 16 * 4 : nullptr;
  if (tl2 && threadIdx.x == 0) tl2[0] = clock64();
  for (int ct = g; ct < Mt; ct += TR_CG) {
    // the NEXT tile's operands from L2 go out before this tile's MFMAs, not behind them: the round trip
    // (~1-2 k clocks) then runs under the MFMAs instead of in front of the next tile
    float afn[NS];
    const bool more = ct + TR_CG < Mt;
    if (more) {
#pragma unroll
      for (int sI = 0; sI < NS; ++sI) afn[sI] = w.trA[(4 * sI + kg) * Mq + 16 * (ct + TR_CG) + ci];
    }
    f32x4 acc[PTW];
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sI = 0; sI < NS; ++sI) {
#pragma unroll
      for (int pt = 0; pt < PTW; ++pt) {
        acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[sI], xf[sI][pt], acc[pt], 0, 0, 0);
      }
    }
    // (the text map of the tile's channels is only needed by the fold: read behind the MFMAs, so that it
    // does not hold registers while they run)
    const float4 tmv = *reinterpret_cast<const float4*>(tm + 16 * ct + 4 * kg);
    const float4 twv = *reinterpret_cast<const float4*>(twl + 16 * ct + 4 * kg);
    const float tm4[4] = {tmv.x, tmv.y, tmv.z, tmv.w};
    const float tw4[4] = {twv.x, twv.y, twv.z, twv.w};
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = tm4[r] * acc[pt][r];
        ssp[pt] += v * v;
        dtp[pt] += tw4[r] * acc[pt][r];
      }
    }
    if (more) {
#pragma unroll
      for (int sI = 0; sI < NS; ++sI) af[sI] = afn[sI];
    }
    if (tl2 && threadIdx.x == 0) tl2[ct == g ? 1 : 2] = clock64();
  }
  // a pixel's channels sit in the four 16-lane rows of the wave: fold them, then across the channel
  // groups.  Two half / row exchanges in registers (v_permlane32_swap, v_permlane16_swap) fold BOTH sums
  // at once -- sum of squares ends up in lanes 0-15, the dot product in lanes 32-47 -- where four
  // __shfl_xor per pixel tile were four ds_bpermute round trips through the LDS hardware.  The four rows
  // are summed as (row0 + row2) + (row1 + row3).
#pragma unroll
  for (int pt = 0; pt < PTW; ++pt) {
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(ssp[pt]), __float_as_uint(dtp[pt]),
                                                    false, false);
    float v = __uint_as_float(h[0]) + __uint_as_float(h[1]);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    if ((kg & 1) == 0 && pt0 + pt < Pt) red[((size_t)g * Pq + 16 * (pt0 + pt) - p_base + ci) * 2 + (kg >> 1)] = v;
  }
  if (tl2 && threadIdx.x == 0) tl2[3] = clock64();
  __syncthreads();
  if (tl && threadIdx.x == 0) tl[2] = clock64();       // debug timeline: MFMA phase done
  const float be = w.be[2][0];
  for (int pl = tid; pl < Pq; pl += NT) {
    const int p = p_base + pl;
    if (p >= HW) continue;
    float s2 = 0.f, d2 = 0.f;
#pragma unroll
    for (int q = 0; q < TR_CG; ++q) {
      s2 += red[((size_t)q * Pq + pl) * 2];
      d2 += red[((size_t)q * Pq + pl) * 2 + 1];
    }
    outp[p] = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;
  }
  __syncthreads();
}

template <int KS>
__device__ __forceinline__ void walk_transform(int tid, const ModuleWeights& w, const WalkArgs& a,
                                               const float* in0, const float* tm, float* outp,
                                               float* scr, long long* tl) {
  // pixel tiles per wave: half of ceil(H*W / 16), at most 6 (H*W <= 192: walk_supported)
  if ((a.H * a.W + 15) / 16 <= 10) walk_transform_t<KS, 5, 8>(tid, w, a, in0, tm, nullptr, outp, scr, tl, 0);
  else walk_transform_t<KS, 6, 8>(tid, w, a, in0, tm, nullptr, outp, scr, tl, 0);
}

// ---------------------------------------------------------------------------------------------
// Text maps of every (step, question) that has a text parameter, hoisted out of the walker:
//   tmap[t, n, :] = word_vecs[t, n, :] . W_txt[ws] + b_txt[ws]     (nmn3_modules.py:53-57,101,161,
//   209,424,479), ws chosen by the token at (t, n).
// One workgroup = up to 8 nodes of ONE weight set at ONE step, so the [E, M] weight stream (300 KB)
// is read once per 8 nodes and the streams of a batch spread over the chip instead of sitting on
// each question's critical path (5.6 us per text node inside the walker).  Table-free: workgroup
// (g, ws, t) scans the step's tokens, ranks the questions whose operator uses weight set ws, and
// takes ranks [8g, 8g + 8); workgroups without work exit.
// ---------------------------------------------------------------------------------------------
constexpr int TMG = 8;
__device__ __forceinline__ int text_ws_of(int op) {
  switch (op) {
    case N2NMN_OP_FIND: case N2NMN_OP_FILTER: return 0;
    case N2NMN_OP_FIND_SAME_PROPERTY: return 1;
    case N2NMN_OP_TRANSFORM: return 2;
    case N2NMN_OP_SAME_PROPERTY: return 3;
    case N2NMN_OP_DESCRIBE: return 4;
    default: return -1;
  }
}

__global__ __launch_bounds__(WT) void walk_textmap_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int wcount[WW];
  __shared__ int sel[TMG];
  const int g = blockIdx.x, ws = blockIdx.y;
  const int t = blockIdx.z % a.T, kb = blockIdx.z / a.T;
  const WalkBatch& B = a.b[kb];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int N = a.N, E = a.E, Mp = a.Mp;
  // ---- rank the questions of this step whose operator reads weight set ws
  int base = 0, total = 0;
  for (int n0 = 0; n0 < N; n0 += WT) {
    const int n = n0 + tid;
    bool hit = false;
    if (n < N) {
      const int tok = B.tokens[(size_t)t * N + n];
      hit = tok >= 0 && tok < a.V && text_ws_of(a.token_op[tok]) == ws;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcount[wid] = __builtin_popcountll(m);
    __syncthreads();
    int before = base;
    for (int q = 0; q < wid; ++q) before += wcount[q];
    int all = 0;
    for (int q = 0; q < WW; ++q) all += wcount[q];
    const int rank = before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (hit && rank >= TMG * g && rank < TMG * g + TMG) sel[rank - TMG * g] = n;
    base += all;
    __syncthreads();
  }
  total = base;
  const int cnt = min(TMG, total - TMG * g);
  if (cnt <= 0) return;
  float* wvl = smem;                         // [TMG][E]
  float* part = wvl + TMG * ((E + 3) & ~3);  // [WW][TMG][256]
  const int Ep = (E + 3) & ~3;
  for (int gi = wid; gi < TMG; gi += WW) {
    const float* src = gi < cnt ? B.word_vecs + ((size_t)t * N + sel[gi]) * E : nullptr;
    for (int e = lane; e < E; e += 64) wvl[gi * Ep + e] = src ? src[e] : 0.f;
  }
  __syncthreads();
  const float* Wp = w.Wtxt[ws];
  const float* bm = w.btxt[ws];
  const int kper = (E + WW - 1) / WW;
  const int k0 = wid * kper, k1 = min(E, k0 + kper);
  constexpr int KU = 19;
  for (int cb = 0; cb < Mp; cb += 256) {
    float4 acc[TMG];
#pragma unroll
    for (int gi = 0; gi < TMG; ++gi) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned col = (unsigned)min(cb + 4 * lane, Mp - 4);
    for (int kq = k0; kq < k1; kq += KU) {
      float4 w4[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const unsigned k = (unsigned)min(kq + u, k1 - 1);
        w4[u] = *reinterpret_cast<const float4*>(Wp + (k * (unsigned)Mp + col));
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (kq + u < k1) {
#pragma unroll
          for (int gi = 0; gi < TMG; ++gi) {
            const float x = wvl[gi * Ep + kq + u];
            acc[gi].x += x * w4[u].x; acc[gi].y += x * w4[u].y; acc[gi].z += x * w4[u].z;
            acc[gi].w += x * w4[u].w;
          }
        }
      }
    }
#pragma unroll
    for (int gi = 0; gi < TMG; ++gi)
      *reinterpret_cast<float4*>(part + ((size_t)(wid * TMG + gi) * 256) + 4 * lane) = acc[gi];
    __syncthreads();
    for (int i = tid; i < TMG * 256; i += WT) {
      const int gi = i >> 8, c = i & 255;
      if (gi < cnt && cb + c < Mp) {
        float r = bm[cb + c];
#pragma unroll
        for (int q = 0; q < WW; ++q) r += part[(size_t)(q * TMG + gi) * 256 + c];
        B.tmap[((size_t)t * N + sel[gi]) * Mp + cb + c] = r;
      }
    }
    __syncthreads();
  }
}

// Find / Filter epilogue for NF nodes of ONE question in one pass over the image's conv_image map:
//   att_f[r] = l2norm_c(M[r,c] * tmap_f[c]) . w_e + b_e     (nmn3_modules.py:104-108; Filter's
// find_result :129).  The nodes share the FindModule weights and the map (153.6 KB) and differ only
// in the text map, so a layout with three Find-type nodes streams the map once instead of three
// times.  One wave per row, all rows of a wave in flight.
template <int CI, int NF>
__device__ __forceinline__ void walk_find_pass(int tid, const ModuleWeights& w, const float* Mbuf,
                                               const float* const* ts, float* const* os, int HW,
                                               int Mp) {
  const int lane = tid & 63, wid = tid >> 6;
  const float be = w.be[0][0];
  float4 t4[NF][CI], e4[CI];
#pragma unroll
  for (int i = 0; i < CI; ++i) {
    const int c = 4 * lane + 256 * i;
    const bool ok = c < Mp;
    e4[i] = ok ? *reinterpret_cast<const float4*>(w.we[0] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NF; ++j)
      t4[j][i] = ok ? *reinterpret_cast<const float4*>(ts[j] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr int UNR = CI == 1 ? 19 : (CI == 2 ? 10 : 5);
  for (int rb = wid; rb < HW; rb += UNR * WW) {
    float4 m4[UNR][CI];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned r = (unsigned)min(rb + u * WW, HW - 1);
#pragma unroll
      for (int i = 0; i < CI; ++i) {
        const unsigned c = (unsigned)min(4 * lane + 256 * i, Mp - 4);
        m4[u][i] = *reinterpret_cast<const float4*>(Mbuf + (r * (unsigned)Mp + c));
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + u * WW;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          const float p0 = m4[u][i].x * t4[j][i].x, p1 = m4[u][i].y * t4[j][i].y,
                      p2 = m4[u][i].z * t4[j][i].z, p3 = m4[u][i].w * t4[j][i].w;
          ss += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
          dot += p0 * e4[i].x + p1 * e4[i].y + p2 * e4[i].z + p3 * e4[i].w;
        }
        const float s2 = wave_sum(ss);
        const float d2 = wave_sum(dot);
        if (lane == 0 && r < HW) os[j][r] = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;
      }
    }
  }
}

// one question's whole module network (walk_kernel's body)
template <int CI>
__device__ __forceinline__ void walk_question(const ModuleWeights& w, const WalkArgs& a, int q, float* smem,
                                              WalkLds& L) {
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int tid0 = threadIdx.x;
  const int H = a.H, W = a.W, HW = H * W, D = a.D, M = a.M, Mp = a.Mp, E = a.E, C = a.C;
  const int HWp = a.HWp, T = a.T;

  float* arena = smem;                                 // [T][HWp]
  float* tml_buf = arena + (size_t)T * HWp;            // [Mp] text map fetched from HBM (word_vecs path)
  float* am0 = tml_buf + Mp;                           // [Mp] fc_att of input 0
  float* am1 = am0 + Mp;                               // [Mp] fc_att of input 1
  float* ev = am1 + Mp;                                // [Mp]
  float* wv = ev + Mp;                                 // [E] word vector
  float* pooled = wv + ((E + 3) & ~3);                 // [2][D]
  float* sa0 = pooled + 2 * (size_t)D;                 // [HWp] softmax weights
  float* sa1 = sa0 + HWp;
  float* rs = sa1 + HWp;                               // [32] reduction scratch
  const int Tep = (a.T_enc + 3) & ~3;
  int* seql = reinterpret_cast<int*>(rs + 32);         // [T_enc] the question's words
  float* tmaps = reinterpret_cast<float*>(seql + Tep); // [T][Mp] text maps of this question's nodes
  float* attall = tmaps + (a.T_enc > 0 ? (size_t)T * Mp : 0);       // [T][T_enc] decoder attention
  float* scr = attall + (a.T_enc > 0 ? (size_t)T * Tep : 0);        // operator scratch

  // ---- 1. decode the RPN column (nmn3_assembler.py:153-222) ---------------------------------
  // the T token loads (and their op-code lookups) go out in parallel; thread 0 then runs the stack
  // machine on LDS values only
  if (tid0 < T) {
    const int tok = B.tokens[(size_t)tid0 * a.N + n];
    L.tok_op[tid0] = (tok < 0 || tok >= a.V) ? -2 : a.token_op[tok];
  }
  __syncthreads();
  if (tid0 == 0) {
    int sp = 0, nn = 0, ok = 1;
    bool has_eos = false;
    for (int t = 0; t < T; ++t) {
      if (L.tok_op[t] == -2) ok = 0;                             // garbage token: not a layout
      if (L.tok_op[t] < 0) has_eos = true;
    }
    if (!has_eos) ok = 0;                                        // 'cannot find <eos>'
    for (int t = 0; ok && t < T; ++t) {
      const int op = L.tok_op[t];
      if (op < 0) break;                                         // <eos>
      int k;
      bool ans;
      switch (op) {
        case N2NMN_OP_SCENE: case N2NMN_OP_FIND: k = 0; ans = false; break;
        case N2NMN_OP_FILTER: case N2NMN_OP_FIND_SAME_PROPERTY: case N2NMN_OP_TRANSFORM:
          k = 1; ans = false; break;
        case N2NMN_OP_AND: case N2NMN_OP_OR: k = 2; ans = false; break;
        case N2NMN_OP_EXIST: case N2NMN_OP_COUNT: case N2NMN_OP_DESCRIBE: k = 1; ans = true; break;
        case N2NMN_OP_EQUAL_NUM: case N2NMN_OP_MORE_NUM: case N2NMN_OP_LESS_NUM:
        case N2NMN_OP_SAME_PROPERTY: k = 2; ans = true; break;
        default: k = -1; ans = false; break;
      }
      if (k < 0 || sp < k) { ok = 0; break; }                    // 'not enough input for ...'
      int i0 = -1, i1 = -1;
      for (int j = k - 1; j >= 0; --j) {                         // input_{k-1} = stack top
        const int top = L.stack[--sp];
        if (L.op[top] & 0x100) { ok = 0; break; }                // 'input incompatible for ...'
        (j == 0 ? i0 : i1) = top;
      }
      if (!ok) break;
      L.op[t] = op | (ans ? 0x100 : 0); L.in0[t] = i0; L.in1[t] = i1;
      L.stack[sp++] = t;
      nn = t + 1;
    }
    if (ok && (sp != 1 || !(L.op[L.stack[0]] & 0x100))) ok = 0;  // stack size / result type
    int nf = 0;
    for (int t = 0; ok && t < nn; ++t) {
      const int o = L.op[t] & 0xff;
      if (o == N2NMN_OP_FIND || o == N2NMN_OP_FILTER) L.flist[nf++] = t;
    }
    L.n_nodes = nn; L.valid = ok; L.n_find = nf;
  }
  __syncthreads();
  float* srow = B.scores + (size_t)n * C;
  if (tid0 == 0 && B.validity) B.validity[n] = L.valid;
  if (tid0 == 0 && a.defer_pool) B.pjob[n] = 0;
  if (!L.valid) {                                                // INVALID_EXPR: zero logits
    for (int c = tid0; c < C; c += WT) srow[c] = 0.f;
    return;
  }
  const int nn = L.n_nodes;
  if (a.stats && tid0 == 0) {
    unsigned long long cf = 0, cpi = 0, cp = 0, ct = 0, ctr = 0;
    for (int t = 0; t < nn; ++t) {
      const int o = L.op[t] & 0xff;
      const bool f = o == N2NMN_OP_FIND || o == N2NMN_OP_FILTER || o == N2NMN_OP_FIND_SAME_PROPERTY;
      const bool p = o == N2NMN_OP_FIND_SAME_PROPERTY || o == N2NMN_OP_SAME_PROPERTY || o == N2NMN_OP_DESCRIBE;
      cf += o == N2NMN_OP_FIND_SAME_PROPERTY; cp += p;      // cf: conv_image map READS (see below)
      cpi += p ? (o == N2NMN_OP_SAME_PROPERTY ? 2 : 1) : 0;
      ct += (f || p || o == N2NMN_OP_TRANSFORM); ctr += o == N2NMN_OP_TRANSFORM;
    }
    // Find / Filter nodes share one pass over the map per 4: here, or in walk_find_kernel ([8])
    const unsigned long long passes = (unsigned long long)((L.n_find + 3) / 4);
    if (a.pre_find) atomicAdd(a.stats + 8, passes); else cf += passes;
    atomicAdd(a.stats + 0, cf); atomicAdd(a.stats + 1, cpi); atomicAdd(a.stats + 2, cp);
    atomicAdd(a.stats + 3, ct); atomicAdd(a.stats + 4, ctr); atomicAdd(a.stats + 5, 1ull);
    if (a.defer_pool && nn > 0) {
      const int ro = L.op[nn - 1] & 0xff;
      if (ro == N2NMN_OP_DESCRIBE || ro == N2NMN_OP_SAME_PROPERTY) {
        atomicAdd(a.stats + 6, 1ull); atomicAdd(a.stats + 7, ro == N2NMN_OP_SAME_PROPERTY ? 2ull : 1ull);
      }
    }
  }
  const float* feat = B.feat + (size_t)n * HW * D;
  int qlen = 0;
  if (a.T_enc > 0) {                                   // the question's words, once
    qlen = min(max(B.seq_len[n], 0), a.T_enc);
    for (int tau = tid0; tau < a.T_enc; tau += WT) {
      const int v = B.seq[(size_t)tau * a.N + n];
      seql[tau] = min(max(v, 0), a.V_txt - 1);
    }
    // ---- text maps of ALL nodes of the question, before the chain starts: they depend only on the
    // decoder's attention.  word_vec = sum_tau att[tau] * emb[word[tau]]  =>  fc_text(word_vec) =
    // b + sum_tau att[tau] * (emb . W_txt)[word[tau]]: a weighted sum of <= len rows of a small
    // L2-resident table.  One WAVE per node (no cross-wave reduction), every node in flight at once.
    for (int i = tid0; i < nn * a.T_enc; i += WT) {
      const int t = i / a.T_enc, tau = i - t * a.T_enc;
      attall[t * Tep + tau] = tau < qlen ? B.atts[((size_t)t * a.T_enc + tau) * a.N + n] : 0.f;
    }
    __syncthreads();
    {
      const int lane0 = tid0 & 63, wid0 = tid0 >> 6;
      for (int t = wid0; t < nn; t += WW) {
        const int o = L.op[t] & 0xff;
        int ws;
        switch (o) {
          case N2NMN_OP_FIND: case N2NMN_OP_FILTER: ws = 0; break;
          case N2NMN_OP_FIND_SAME_PROPERTY: ws = 1; break;
          case N2NMN_OP_TRANSFORM: ws = 2; break;
          case N2NMN_OP_SAME_PROPERTY: ws = 3; break;
          case N2NMN_OP_DESCRIBE: ws = 4; break;
          default: ws = -1; break;
        }
        if (ws < 0) continue;                            // wave-uniform
        const float* ewp = a.ew[ws];
        const float* at = attall + t * Tep;
        for (int cb = 0; cb < Mp; cb += 256) {
          const unsigned col = (unsigned)min(cb + 4 * lane0, Mp - 4);
          float4 acc = *reinterpret_cast<const float4*>(w.btxt[ws] + col);
          constexpr int RU = 16;
          for (int tb = 0; tb < qlen; tb += RU) {
            float4 r4[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
              const int tau = min(tb + u, a.T_enc - 1);
              r4[u] = *reinterpret_cast<const float4*>(ewp + ((unsigned)seql[tau] * (unsigned)Mp + col));
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
              const float av = tb + u < qlen ? at[tb + u] : 0.f;
              acc.x += av * r4[u].x; acc.y += av * r4[u].y; acc.z += av * r4[u].z; acc.w += av * r4[u].w;
            }
          }
          if (cb + 4 * lane0 < Mp) *reinterpret_cast<float4*>(tmaps + (size_t)t * Mp + col) = acc;
        }
      }
    }
    __syncthreads();
  }
  const int ncol = D / 4, nrow = WT / ncol;
  constexpr int PR = WALK_POOL_ROWS;

  // ---- Find / Filter epilogues of the whole question in ONE pass over the image's conv_image map
  // (walk_find_pass), off the dependent chain
  if (a.pre_find) {
    // walk_find_kernel streamed the maps chip-wide: the logits of this question's Find / Filter nodes
    // are 600 bytes each in HBM
    const float* src = B.watt + (size_t)n * T * HWp;
    for (int i = tid0; i < L.n_find * HWp; i += WT) {
      const int t = L.flist[i / HWp], r = i % HWp;
      arena[(size_t)t * HWp + r] = src[(size_t)t * HWp + r];
    }
    __syncthreads();
  } else {
    const float* Mbuf = B.mfind + (size_t)n * HW * Mp;
    for (int f0 = 0; f0 < L.n_find; f0 += 4) {
      const int nf = min(4, L.n_find - f0);
      const float* ts[4];
      float* os[4];
      for (int j = 0; j < 4; ++j) {
        const int t = L.flist[f0 + min(j, nf - 1)];
        ts[j] = a.T_enc > 0 ? tmaps + (size_t)t * Mp : B.tmap + ((size_t)t * a.N + n) * Mp;
        os[j] = arena + (size_t)t * HWp;
      }
      switch (nf) {
        case 1: walk_find_pass<CI, 1>(tid0, w, Mbuf, ts, os, HW, Mp); break;
        case 2: walk_find_pass<CI, 2>(tid0, w, Mbuf, ts, os, HW, Mp); break;
        case 3: walk_find_pass<CI, 3>(tid0, w, Mbuf, ts, os, HW, Mp); break;
        default: walk_find_pass<CI, 4>(tid0, w, Mbuf, ts, os, HW, Mp); break;
      }
    }
    __syncthreads();
  }

  // ---- 2. nodes in token order -------------------------------------------------------------
  for (int t = 0; t < nn; ++t) {
    // every per-thread index below derives from this opaque copy, so the compiler cannot hoist the
    // (dozens of) per-load address computations of one operator out of the node loop, where they
    // would stay live across all the other operators and spill
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wid = tid >> 6;
    const int lc = tid % ncol, lr = tid / ncol;
    const int op = L.op[t] & 0xff;
    float* tml = tml_buf;
    long long* tl = a.timeline ? a.timeline + ((size_t)q * MAXT + t) * 4 : nullptr;
    if (tl && tid0 == 0) tl[0] = clock64();
    const float* in0 = L.in0[t] >= 0 ? arena + (size_t)L.in0[t] * HWp : nullptr;
    const float* in1 = L.in1[t] >= 0 ? arena + (size_t)L.in1[t] * HWp : nullptr;
    float* outp = arena + (size_t)t * HWp;

    const bool pools = op == N2NMN_OP_FIND_SAME_PROPERTY || op == N2NMN_OP_SAME_PROPERTY ||
                       op == N2NMN_OP_DESCRIBE;

    // text map of this node (fc_text / text_fc of its text parameter, nmn3_modules.py:53-57,101,
    // 161,209,424,479)
    const bool has_text = op == N2NMN_OP_FIND || op == N2NMN_OP_FILTER || pools ||
                          op == N2NMN_OP_TRANSFORM;
    if (has_text && a.T_enc > 0) {
      tml = tmaps + (size_t)t * Mp;                      // computed in the pre-pass above
    } else if (has_text) {                              // hoisted: walk_textmap_kernel's row
      const float* src = B.tmap + ((size_t)t * a.N + n) * Mp;
      for (int c = 4 * tid; c < Mp; c += 4 * WT)
        *reinterpret_cast<float4*>(tml + c) = *reinterpret_cast<const float4*>(src + c);
    }
    if (tl && tid0 == 0 && op != N2NMN_OP_TRANSFORM) tl[1] = clock64();
    // throughput mode: the answer operators that pool (Describe / SameProperty, always the root) only
    // compute their soft-max weights here; the feature stream, fc_att and the answer head of ALL
    // such questions of the launch run chip-wide in walk_pool_kernel / walk_heads_kernel
    const bool defer = a.defer_pool && pools && op != N2NMN_OP_FIND_SAME_PROPERTY;
    if (defer) {
      const int nin = op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
      float* pw = B.pw + (size_t)n * 2 * HWp;
      for (int i = 0; i < nin; ++i) {                  // :432-437,482-484
        const float* src = i == 0 ? in0 : in1;
        float lm = -INFINITY;
        for (int r = tid; r < HW; r += WT) lm = fmaxf(lm, src[r]);
        const float mx = wg_reduce<1>(lm, rs);
        float ls = 0.f;
        for (int r = tid; r < HW; r += WT) ls += expf(src[r] - mx);
        const float sum = wg_reduce<0>(ls, rs);
        for (int r = tid; r < HW; r += WT) pw[i * HWp + r] = expf(src[r] - mx) / sum;
      }
      float* ptm = B.ptm + (size_t)n * Mp;
      for (int c = tid; c < Mp; c += WT) ptm[c] = tml[c];
      if (tid == 0) {
        B.pjob[n] = op;
        if (a.plist) {                                 // the job lists walk_fcatt_kernel works from
          const int ty = op == N2NMN_OP_SAME_PROPERTY ? 1 : 0;
          const int j = atomicAdd(a.cnt + 3 + ty, 1);
          if (j < a.pcap) a.plist[ty * a.pcap + j] = q;
        }
      }
      break;                                           // the root is the last node
    }
    if (pools) {
      // the feature rows of this thread do not depend on the softmax: all of its 16-B loads go out
      // first (the whole [HW, D] map of the question is in flight at once) and land while the
      // softmax is computed
      float4 fr[PR];
      const int myrows = lr < nrow ? (HW - lr + nrow - 1) / nrow : 0;
      {
        const unsigned rowstep = (unsigned)(nrow * D);
        unsigned off = (unsigned)(lr * D + 4 * lc);
        const unsigned last = (unsigned)(((myrows > 0 ? lr + (myrows - 1) * nrow : 0)) * D + 4 * lc);
#pragma unroll
        for (int qq = 0; qq < PR; ++qq) {
          fr[qq] = *reinterpret_cast<const float4*>(feat + min(off, last));
          off += rowstep;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // softmax over the H*W logits of each input (:170-172,432-437,482-484)
      const int nin = op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
      for (int i = 0; i < nin; ++i) {
        const float* src = i == 0 ? in0 : in1;
        float* dst = i == 0 ? sa0 : sa1;
        float lm = -INFINITY;
        for (int r = tid; r < HW; r += WT) lm = fmaxf(lm, src[r]);
        const float mx = wg_reduce<1>(lm, rs);
        float ls = 0.f;
        for (int r = tid; r < HW; r += WT) {
          const float ex = expf(src[r] - mx);
          dst[r] = ex;
          ls += ex;
        }
        const float sum = wg_reduce<0>(ls, rs);
        for (int r = tid; r < HW; r += WT) dst[r] = dst[r] / sum;
      }
      __syncthreads();
      float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
      float* stage = scr;                                // [nrow][2][D]
      if (lr < nrow) {
#pragma unroll
        for (int qq = 0; qq < PR; ++qq) {
          if (qq < myrows) {
            const int r = lr + qq * nrow;
            const float w0 = sa0[r], w1 = nin == 2 ? sa1[r] : 0.f;
            acc0.x += w0 * fr[qq].x; acc0.y += w0 * fr[qq].y; acc0.z += w0 * fr[qq].z; acc0.w += w0 * fr[qq].w;
            acc1.x += w1 * fr[qq].x; acc1.y += w1 * fr[qq].y; acc1.z += w1 * fr[qq].z; acc1.w += w1 * fr[qq].w;
          }
        }
        *reinterpret_cast<float4*>(stage + (size_t)(lr * 2 + 0) * D + 4 * lc) = acc0;
        *reinterpret_cast<float4*>(stage + (size_t)(lr * 2 + 1) * D + 4 * lc) = acc1;
      }
      __syncthreads();
      for (int i = tid; i < nin * D; i += WT) {
        const int which = i / D, c = i - which * D;
        float s = 0.f;
        for (int qq = 0; qq < nrow; ++qq) s += stage[(size_t)(qq * 2 + which) * D + c];
        pooled[i] = s;
      }
      __syncthreads();
      // fc_att (:173-176,438-446,487-490)
      const int wi0 = op == N2NMN_OP_FIND_SAME_PROPERTY ? 0 : (op == N2NMN_OP_SAME_PROPERTY ? 1 : 3);
      fc_pad<32>(tid, pooled, D, w.Watt[wi0], w.batt[wi0], Mp, am0, scr, nullptr);
      if (nin == 2) fc_pad<32>(tid, pooled + D, D, w.Watt[2], w.batt[2], Mp, am1, scr, nullptr);
    }

    __syncthreads();
    if (tl && tid0 == 0 && op != N2NMN_OP_TRANSFORM) tl[2] = clock64();
    switch (op) {
      case N2NMN_OP_SCENE:                                       // :60-72
        for (int r = tid; r < HW; r += WT) outp[r] = 3.0f;
        break;
      case N2NMN_OP_AND:                                         // :218-236
        for (int r = tid; r < HW; r += WT) outp[r] = fminf(in0[r], in1[r]);
        break;
      case N2NMN_OP_OR:                                          // :238-256
        for (int r = tid; r < HW; r += WT) outp[r] = fmaxf(in0[r], in1[r]);
        break;
      case N2NMN_OP_FIND:                                        // done in the pre-pass
        break;
      case N2NMN_OP_FILTER:                                      // And(input_0, find_result) :129-130
        for (int r = tid; r < HW; r += WT) outp[r] = fminf(in0[r], outp[r]);
        break;
      case N2NMN_OP_FIND_SAME_PROPERTY: {
        // att[r] = l2norm_c(M[r,c] * tmap[c] (* amap[c])) . w_e + b_e  [min with input_0: Filter]
        // (:104-108, :129-130, :178-180); one wave per row, all rows of a wave in flight
        constexpr bool fsp = true;
        const int wsel = fsp ? 1 : 0;
        const float* Mbuf = (fsp ? B.mfsp : B.mfind) + (size_t)n * HW * Mp;
        const float be = w.be[wsel][0];
        float4 t4[CI], e4[CI];
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          const int c = 4 * lane + 256 * i;
          if (c < Mp) {
            t4[i] = *reinterpret_cast<const float4*>(tml + c);
            e4[i] = *reinterpret_cast<const float4*>(w.we[wsel] + c);
            if (fsp) {
              const float4 a4 = *reinterpret_cast<const float4*>(am0 + c);
              t4[i].x *= a4.x; t4[i].y *= a4.y; t4[i].z *= a4.z; t4[i].w *= a4.w;
            }
          } else {
            t4[i] = make_float4(0.f, 0.f, 0.f, 0.f); e4[i] = t4[i];
          }
        }
        constexpr int UNR = CI == 1 ? 19 : (CI == 2 ? 10 : 5);
        for (int rb = wid; rb < HW; rb += UNR * WW) {
          float4 m4[UNR][CI];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const unsigned r = (unsigned)min(rb + u * WW, HW - 1);
#pragma unroll
            for (int i = 0; i < CI; ++i) {
              const unsigned c = (unsigned)min(4 * lane + 256 * i, Mp - 4);
              m4[u][i] = *reinterpret_cast<const float4*>(Mbuf + (r * (unsigned)Mp + c));
            }
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int r = rb + u * WW;
            float ss = 0.f, dot = 0.f;
#pragma unroll
            for (int i = 0; i < CI; ++i) {
              const float p0 = m4[u][i].x * t4[i].x, p1 = m4[u][i].y * t4[i].y,
                          p2 = m4[u][i].z * t4[i].z, p3 = m4[u][i].w * t4[i].w;
              ss += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
              dot += p0 * e4[i].x + p1 * e4[i].y + p2 * e4[i].z + p3 * e4[i].w;
            }
            const float s2 = wave_sum(ss);
            const float d2 = wave_sum(dot);
            if (lane == 0 && r < HW) {
              float att = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;     // tf.nn.l2_normalize eps (A.4)
              outp[r] = att;
            }
          }
        }
        break;
      }
      case N2NMN_OP_TRANSFORM:                                   // :185-216
        if (a.ksize == 5) walk_transform<5>(tid, w, a, in0, tml, outp, scr, tl);
        else walk_transform<3>(tid, w, a, in0, tml, outp, scr, tl);
        break;
      case N2NMN_OP_DESCRIBE:                                    // :479-493
      case N2NMN_OP_SAME_PROPERTY: {                             // :424-450
        const bool same = op == N2NMN_OP_SAME_PROPERTY;
        float lss = 0.f;
        for (int c = tid; c < Mp; c += WT) {
          float v = 0.f;
          if (c < M) {
            v = am0[c] * tml[c];
            if (same) v *= am1[c];
          }
          ev[c] = v;
          lss += v * v;
        }
        const float ss = wg_reduce<0>(lss, rs);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int c = tid; c < Mp; c += WT) ev[c] *= inv;
        __syncthreads();
        const int wi = same ? 5 : 6;
        fc_out(tid, ev, M, w.Wans[wi], w.bans[wi], C, srow, scr);
        break;
      }
      default: {
        // Exist (:258-280), Count (:282-304), EqualNum / MoreNum / LessNum (:306-400)
        float* x = scr;                                  // up to 2*HW + 4 features
        float* red = x + ((2 * HW + 4 + 3) & ~3);
        const int nin = (op == N2NMN_OP_EXIST || op == N2NMN_OP_COUNT) ? 1 : 2;
        float mn[2], mx[2], sm[2];
        for (int i = 0; i < nin; ++i) {
          const float* src = i == 0 ? in0 : in1;
          float lmn = INFINITY, lmx = -INFINITY, lsm = 0.f;
          for (int r = tid; r < HW; r += WT) {
            const float v = src[r];
            x[i * (HW + 2) + r] = v;                     // row-major y*W + x flatten (:297)
            lmn = fminf(lmn, v); lmx = fmaxf(lmx, v); lsm += v;
          }
          mn[i] = wg_reduce<2>(lmn, rs);
          mx[i] = wg_reduce<1>(lmx, rs);
          sm[i] = wg_reduce<0>(lsm, rs);
        }
        __syncthreads();
        int F, wi;
        if (op == N2NMN_OP_EXIST) {
          if (tid == 0) { x[0] = mn[0]; x[1] = sm[0] / (float)HW; x[2] = mx[0]; }
          F = 3; wi = 0;
        } else if (op == N2NMN_OP_COUNT) {
          if (tid == 0) { x[HW] = mn[0]; x[HW + 1] = mx[0]; }
          F = HW + 2; wi = 1;
        } else {
          if (tid == 0) {
            x[HW] = mn[0]; x[HW + 1] = mx[0];
            x[2 * HW + 2] = mn[1]; x[2 * HW + 3] = mx[1];
          }
          F = 2 * HW + 4;
          wi = op == N2NMN_OP_EQUAL_NUM ? 2 : (op == N2NMN_OP_MORE_NUM ? 3 : 4);
        }
        __syncthreads();
        fc_out(tid, x, F, w.Wans[wi], w.bans[wi], C, srow, red);
        break;
      }
    }
    __syncthreads();
    if (tl && tid0 == 0) tl[3] = clock64();
  }
}

// Plain passes: workgroup = question.  Staged passes: only the questions the plan listed as nested deeper
// than the pass launches levels for (fblist, its length on the device) -- a persistent grid of 64
// workgroups walks the list, so the template mix (an empty list) pays for 64 workgroups that leave at
// once instead of one 150 KB-LDS workgroup per question of the pass (4.7 us).
template <int CI>
__global__ __launch_bounds__(WT) void walk_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ WalkLds L;
  if (!a.staged) {
    walk_question<CI>(w, a, blockIdx.x, smem, L);
    return;
  }
  const int nfb = min(a.cnt[1], a.K * a.N);
  for (int i = blockIdx.x; i < nfb; i += gridDim.x) {
    __syncthreads();                                   // the previous question's readers of L / smem are done
    if (a.stats && threadIdx.x == 0) atomicAdd(a.stats + 9, 1ull);
    walk_question<CI>(w, a, a.fblist[i], smem, L);
  }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------
// Chip-wide front end of the walker for passes of many questions (throughput mode): the two things
// every question does that do NOT depend on its tree -- the text maps of its nodes and the Find /
// Filter epilogues over the image's conv_image map -- leave the one-workgroup-per-question chain
// (where they ran at one CU's pace: 13 % of HBM) and run as launches whose workgroups
// fill the chip.
//
// walk_tmap_kernel: text maps of all nodes of a question from the decoder's attention and the
// commit-time (embedding . W_txt) tables (see walk_kernel's pre-pass), one wave per node, written to
// tmap[t][n][:].  One workgroup (4 waves) per question.
// walk_find_kernel: att_f[r] = l2norm_c(M[r,c] * tmap_f[c]) . w_e + b_e (nmn3_modules.py:104-108,
// Filter's find_result :129) for the Find / Filter nodes of a question, WALK_FIND_PARTS workgroups per
// question, each streaming its rows of the map once for up to 4 nodes; logits to watt[n][t][:].
// Both re-read the token column (20 ints) instead of sharing a decoded table: a Find-type node is a
// token before the first <eos> whose operator is Find / Filter; for an invalid layout the rows they
// write are never read.
// ---------------------------------------------------------------------------------------------
constexpr int FT = 256, FW = FT / 64;

// Reverse-Polish decode of one layout by ONE thread (nmn3_assembler.py:153-222, the walker's stack
// machine: same checks in the same order), plus what the staged walker needs per node: heavy depth
// and subtree start.  tok_op[t] = op code of token t, -1 for <eos>, -2 for a token out of range.
// The loop is a chain of dependent LDS accesses, so it is built to need ONE LDS round trip per node:
// a node's attributes are one packed word (op | answer << 7 | hd << 8 | lo << 16), the two topmost stack
// entries live in registers, and the stack proper is only read back behind a two-input operator.
__device__ inline int plan_layout(const int* tok_op, int T, WalkProg& P, int* stack, unsigned* word) {
  int nn = 0, ok = 1;
  bool has_eos = false;
  for (int t = 0; t < T; ++t) {
    if (tok_op[t] == -2) ok = 0;
    if (tok_op[t] < 0) has_eos = true;
  }
  if (!has_eos) ok = 0;
  int maxhd = 0, nf = 0;
  int sp = 0, a = -1, b = -1;              // stack depth; a = top, b = second (stack[0 .. sp - 3] in LDS)
  for (int t = 0; ok && t < T; ++t) {
    const int op = tok_op[t];
    if (op < 0) break;
    int k;
    bool ans;
    switch (op) {
      case N2NMN_OP_SCENE: case N2NMN_OP_FIND: k = 0; ans = false; break;
      case N2NMN_OP_FILTER: case N2NMN_OP_FIND_SAME_PROPERTY: case N2NMN_OP_TRANSFORM:
        k = 1; ans = false; break;
      case N2NMN_OP_AND: case N2NMN_OP_OR: k = 2; ans = false; break;
      case N2NMN_OP_EXIST: case N2NMN_OP_COUNT: case N2NMN_OP_DESCRIBE: k = 1; ans = true; break;
      case N2NMN_OP_EQUAL_NUM: case N2NMN_OP_MORE_NUM: case N2NMN_OP_LESS_NUM:
      case N2NMN_OP_SAME_PROPERTY: k = 2; ans = true; break;
      default: k = -1; ans = false; break;
    }
    if (k < 0 || sp < k) { ok = 0; break; }                      // 'not enough input for ...'
    int i0 = -1, i1 = -1;
    unsigned w0 = 0, w1 = 0;
    if (k == 1) { i0 = a; w0 = word[a]; }
    else if (k == 2) { i1 = a; i0 = b; w1 = word[a]; w0 = word[b]; }
    if ((w0 | w1) & 0x80u) { ok = 0; break; }                    // 'input incompatible for ...'
    const bool heavy = op == N2NMN_OP_TRANSFORM || op == N2NMN_OP_FIND_SAME_PROPERTY;
    int hd = heavy ? 1 : 0, lo = t;
    if (i0 >= 0) { hd += (int)((w0 >> 8) & 0xff); lo = (int)((w0 >> 16) & 0xff); }
    if (i1 >= 0) hd = max(hd, (int)((w1 >> 8) & 0xff) + (heavy ? 1 : 0));
    word[t] = (unsigned)op | (ans ? 0x80u : 0u) | ((unsigned)hd << 8) | ((unsigned)lo << 16);
    P.op[t] = (uint8_t)(op | (ans ? 0x80 : 0)); P.in0[t] = (int8_t)i0; P.in1[t] = (int8_t)i1;
    P.hd[t] = (uint8_t)hd; P.lo[t] = (uint8_t)lo;
    if (op == N2NMN_OP_FIND || op == N2NMN_OP_FILTER) P.flist[nf++] = (uint8_t)t;
    maxhd = max(maxhd, hd);
    // pop k, push t
    if (k == 0) { if (sp >= 2) stack[sp - 2] = b; b = a; a = t; }
    else if (k == 1) { a = t; }
    else { a = t; b = sp >= 3 ? stack[sp - 3] : -1; }
    sp += 1 - k;
    nn = t + 1;
  }
  if (ok && (sp != 1 || !(word[a] & 0x80u))) ok = 0;             // stack size / result type
  P.nn = nn; P.valid = ok; P.nfind = nf;
  P.fallback = ok && maxhd >= 2;          // (the caller decides: nesting deeper than the levels it lists)
  return ok ? maxhd : 0;
}

__global__ __launch_bounds__(FT) void walk_tmap_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int ops[MAXT];
  __shared__ int s_nn;
  __shared__ __attribute__((aligned(16))) WalkProg P;
  __shared__ int stack[MAXT];
  __shared__ unsigned word[MAXT];
  const int q = blockIdx.x;
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = a.T, Mp = a.Mp, Te = a.T_enc;
  const int Tep = (Te + 3) & ~3;
  int* seql = reinterpret_cast<int*>(smem);                 // [Tep]
  float* attall = smem + Tep;                               // [T][Tep]
  if (tid < T) {
    const int tok = B.tokens[(size_t)tid * a.N + n];
    ops[tid] = (tok < 0 || tok >= a.V) ? -2 : a.token_op[tok];
  }
  __syncthreads();
  if (tid == 0) {
    int nn = 0;
    while (nn < T && ops[nn] >= 0) ++nn;
    s_nn = nn;
    // the pass's plan: every layout is decoded once, here (walk_find / walk_heavy / walk_light read it)
    if (a.pre_find) {
      const int deepest = plan_layout(ops, T, P, stack, word);
      // nesting of Transform / FindSameProperty up to the levels this pass launches is listed level by
      // level; deeper layouts go to the one-workgroup walker (and the host, told through cnt[11], launches
      // more levels for the passes that follow)
      P.fallback = P.valid && deepest > max(a.hlevels, 1);
      // (n2nmn_walk_set_nesting_bound: the host vouched that this cannot happen and launches no fall-back
      // walker; a layout that breaks the promise is reported invalid -- zero logits -- not served wrongly)
      if (a.no_fallback && P.fallback) { P.valid = 0; P.fallback = 0; }
      if (a.staged && deepest >= 2) atomicMax(a.cnt + 5, deepest);
    }
  }
  const int qlen = min(max(B.seq_len[n], 0), Te);
  for (int tau = tid; tau < Te; tau += FT) {
    const int v = B.seq[(size_t)tau * a.N + n];
    seql[tau] = min(max(v, 0), a.V_txt - 1);
  }
  __syncthreads();
  if (a.pre_find) {                                         // 208 bytes per question
    const int4* src = reinterpret_cast<const int4*>(&P);
    int4* dst = reinterpret_cast<int4*>(B.prog + n);
    if (tid < (int)(sizeof(WalkProg) / 16)) dst[tid] = src[tid];
  }
  const int nn = s_nn;
  for (int i = tid; i < nn * Te; i += FT) {
    const int t = i / Te, tau = i - t * Te;
    attall[t * Tep + tau] = tau < qlen ? B.atts[((size_t)t * Te + tau) * a.N + n] : 0.f;
  }
  __syncthreads();
  for (int t = wid; t < nn; t += FW) {
    const int ws = text_ws_of(ops[t]);
    if (ws < 0) continue;                                    // wave-uniform
    const float* ewp = a.ew[ws];
    const float* at = attall + t * Tep;
    for (int cb = 0; cb < Mp; cb += 256) {
      const unsigned col = (unsigned)min(cb + 4 * lane, Mp - 4);
      float4 acc = *reinterpret_cast<const float4*>(w.btxt[ws] + col);
      constexpr int RU = 16;
      for (int tb = 0; tb < qlen; tb += RU) {
        float4 r4[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int tau = min(tb + u, Te - 1);
          r4[u] = *reinterpret_cast<const float4*>(ewp + ((unsigned)seql[tau] * (unsigned)Mp + col));
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const float av = tb + u < qlen ? at[tb + u] : 0.f;
          acc.x += av * r4[u].x; acc.y += av * r4[u].y; acc.z += av * r4[u].z; acc.w += av * r4[u].w;
        }
      }
      if (cb + 4 * lane < Mp)
        *reinterpret_cast<float4*>(B.tmap + ((size_t)t * a.N + n) * Mp + col) = acc;
    }
  }
  // the chip-wide job lists of the staged walker (WalkArgs::staged) -- last, so that the returning
  // atomics (an L2 round trip each) delay nothing: one list per (nesting level, operator); a level's
  // FindSameProperty jobs are handed out first by walk_heavy_kernel
  if (a.staged && tid == 0 && P.valid) {
    if (P.fallback) {
      a.fblist[atomicAdd(a.cnt + 1, 1)] = q;
    } else {
      for (int t = 0; t < P.nn; ++t) {
        const int o = P.op[t] & 0x7f;
        const int kind = o == N2NMN_OP_TRANSFORM ? 0 : (o == N2NMN_OP_FIND_SAME_PROPERTY ? 1 : -1);
        const int lv = (int)P.hd[t] - 1;
        if (kind < 0 || lv < 0 || lv >= WALK_HLEVELS) continue;
        const int cap = (a.hoff[lv + 1] - a.hoff[lv]) / 2;
        const int j = atomicAdd(a.cnt + (lv == 0 ? 2 * kind : 6 + 2 * lv + kind), 1);
        if (j < cap) a.hjobs[a.hoff[lv] + kind * cap + j] = (q << 8) | t;
      }
    }
  }
}

// rows of a wave in flight per batch (CI float4 column groups each)
template <int CI> struct FindUnroll { static constexpr int value = CI == 1 ? 5 : (CI == 2 ? 5 : 3); };

// the map rows [rb, rb + UNR * FW) of this wave (row = rb + u * FW), 16 bytes per lane and column group
template <int CI> struct FindRows { float4 v[FindUnroll<CI>::value][CI]; };
template <int CI>
__device__ __forceinline__ FindRows<CI> find_load(const float* Mbuf, int rb, int r1, int lane, int Mp) {
  constexpr int UNR = FindUnroll<CI>::value;
  FindRows<CI> m;
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const unsigned r = (unsigned)min(rb + u * FW, r1 - 1);
#pragma unroll
    for (int i = 0; i < CI; ++i) {
      const unsigned c = (unsigned)min(4 * lane + 256 * i, Mp - 4);
      m.v[u][i] = *reinterpret_cast<const float4*>(Mbuf + (r * (unsigned)Mp + c));
    }
  }
  return m;
}

// `pre` holds the first batch of this wave's rows (rows r0 + wid + u * FW), requested by the caller
// before anything else.  t4 / e4: text maps of the NF nodes and conv_eltwise weights at this lane's columns.
template <int CI, int NF, bool ONE>
__device__ __forceinline__ void find_rows_core(int tid, float be, const float4 (&e4)[CI],
                                               const float4 (&t4)[NF][CI], const float* Mbuf,
                                               float* const* os, int r0, int r1, int Mp,
                                               const FindRows<CI> pre) {
  const int lane = tid & 63, wid = tid >> 6;
  constexpr int UNR = FindUnroll<CI>::value;
  for (int rb = r0 + wid; rb < r1; rb += UNR * FW) {
    // ONE: the first batch is all of this wave's rows (r1 - r0 <= UNR * FW, checked by the launcher)
    const FindRows<CI> m = (ONE || rb == r0 + wid) ? pre : find_load<CI>(Mbuf, rb, r1, lane, Mp);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + u * FW;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          const float p0 = m.v[u][i].x * t4[j][i].x, p1 = m.v[u][i].y * t4[j][i].y,
                      p2 = m.v[u][i].z * t4[j][i].z, p3 = m.v[u][i].w * t4[j][i].w;
          ss += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
          dot += p0 * e4[i].x + p1 * e4[i].y + p2 * e4[i].z + p3 * e4[i].w;
        }
        float s2, d2;
        wave_sum2(ss, dot, s2, d2);              // both 64-lane sums in one DPP sequence
        if (lane == 0 && r < r1) os[j][r] = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;
      }
    }
  }
}

template <int CI, int NF, bool ONE>
__device__ __forceinline__ void find_rows(int tid, const ModuleWeights& w, const float* Mbuf,
                                          const float* const* ts, float* const* os, int r0, int r1,
                                          int Mp, const FindRows<CI> pre) {
  const int lane = tid & 63;
  float4 t4[NF][CI], e4[CI];
#pragma unroll
  for (int i = 0; i < CI; ++i) {
    const int c = 4 * lane + 256 * i;
    const bool ok = c < Mp;
    e4[i] = ok ? *reinterpret_cast<const float4*>(w.we[0] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NF; ++j)
      t4[j][i] = ok ? *reinterpret_cast<const float4*>(ts[j] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  find_rows_core<CI, NF, ONE>(tid, w.be[0][0], e4, t4, Mbuf, os, r0, r1, Mp, pre);
}

// The question's layout comes from the pass's plan (WalkProg written by walk_tmap_kernel): a scalar
// load of the Find / Filter node list instead of the chain tokens -> op codes -> scan behind two
// barriers, which every one of the 4 x questions workgroups used to run IN FRONT of its map stream.
// The rows of the image's conv_image map do not depend on the layout at all: their loads go out first.
template <int CI, bool ONE>
__global__ __launch_bounds__(FT, (CI == 1 ? 8 : 1)) void walk_find_kernel(ModuleWeights w, WalkArgs a) {
  const int q = blockIdx.x;
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int tid = threadIdx.x;
  const int T = a.T, HW = a.H * a.W, Mp = a.Mp, HWp = a.HWp;
  const int rpp = (HW + WALK_FIND_PARTS - 1) / WALK_FIND_PARTS;
  const int r0 = blockIdx.y * rpp, r1 = min(HW, r0 + rpp);
  if (r0 >= r1) return;
  const float* Mbuf = B.mfind + (size_t)n * HW * Mp;
  const WalkProg* P = B.prog + n;                    // uniform address: scalar loads
  const int nfind = P->valid ? P->nfind : 0;
  const FindRows<CI> pre = find_load<CI>(Mbuf, r0 + (tid >> 6), r1, tid & 63, Mp);
  if (nfind == 0) return;
  // two nodes per sweep over the rows (they sit in registers): the live state of a sweep -- two text maps,
  // two (sum of squares, dot) pairs per row -- leaves room for five waves per SIMD without spills; four
  // nodes per sweep took 127 VGPRs
  for (int f0 = 0; f0 < nfind; f0 += 2) {
    const int nf = min(2, nfind - f0);
    const float* ts[2];
    float* os[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = P->flist[f0 + min(j, nf - 1)];
      ts[j] = B.tmap + ((size_t)t * a.N + n) * Mp;
      os[j] = B.watt + ((size_t)n * T + t) * HWp;
    }
    if (nf == 1) find_rows<CI, 1, ONE>(tid, w, Mbuf, ts, os, r0, r1, Mp, pre);
    else find_rows<CI, 2, ONE>(tid, w, Mbuf, ts, os, r0, r1, Mp, pre);
  }
}

// walk_find16_kernel (round 5): the same epilogue with SIXTEEN lanes per map row instead of sixty-four.
// walk_find_kernel gives a row to a whole wave (a float4 per lane at Mp = 256), so every (row, node) pair
// costs a 64-lane reduction of two values (8 DPP steps + 2 v_readlane) and an IEEE sqrt + division issued
// for ONE useful value: ~40 VALU instructions around 12 FMAs.  Here a wave holds 4 rows at a time (lane =
// (row of the group, 16 channels as NL float4 at 64-B stride: the 16 lanes of a row read 256 contiguous
// bytes per load), the reduction stays inside the 16-lane DPP row (4 steps, no cross-row traffic), and the
// square root / division serve four rows per instruction: 2.1 x fewer VALU instructions per map byte.
// A workgroup (4 waves) owns 32 rows -- 8 per wave, all in flight before the plan is read -- and sweeps the
// question's Find / Filter nodes one at a time over rows that stay in registers (the next node's text map
// is requested under the current sweep).  Measured and not kept: the same without that prefetch (79 VGPRs, six
// waves per SIMD: 26.0 vs 26.7 us warm, equal cold) and one workgroup per question walking its map in chunks
// with a register double buffer (38.7 vs 36.7 us cold).
constexpr int F16_ROWS = 32;
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xF>(0.f, v);    // row_half_mirror
  v += dpp_f<0x140, 0xF>(0.f, v);    // row_mirror: every lane holds its 16-lane row's sum
  return v;
}
template <int NL>
__global__ __launch_bounds__(FT, 5) void walk_find16_kernel(ModuleWeights w, WalkArgs a) {
  const int q = blockIdx.x;
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int cl = lane & 15, rg = lane >> 4;
  const int T = a.T, HW = a.H * a.W, Mp = a.Mp, HWp = a.HWp;
  const int r0 = blockIdx.y * F16_ROWS + 8 * wid;      // this wave's rows r0 .. r0 + 7 (wave-uniform)
  if (r0 >= HW) return;
  const float* Mbuf = B.mfind + (size_t)n * HW * Mp;
  float4 m[2][NL];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const unsigned r = (unsigned)min(r0 + 4 * u + rg, HW - 1);
#pragma unroll
    for (int i = 0; i < NL; ++i)
      m[u][i] = *reinterpret_cast<const float4*>(Mbuf + (r * (unsigned)Mp + 4u * cl + 64u * i));
  }
  const WalkProg* P = B.prog + n;                      // uniform address: scalar loads
  const int nfind = P->valid ? P->nfind : 0;
  if (nfind == 0) return;
  const float be = w.be[0][0];
  float4 e4[NL], tn[NL];
  {
    const float* t0 = B.tmap + ((size_t)P->flist[0] * a.N + n) * Mp;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      e4[i] = *reinterpret_cast<const float4*>(w.we[0] + 4 * cl + 64 * i);
      tn[i] = *reinterpret_cast<const float4*>(t0 + 4 * cl + 64 * i);
    }
  }
  for (int f = 0; f < nfind; ++f) {
    const int t = P->flist[f];
    float4 t4[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) t4[i] = tn[i];
    if (f + 1 < nfind) {
      const float* t1 = B.tmap + ((size_t)P->flist[f + 1] * a.N + n) * Mp;
#pragma unroll
      for (int i = 0; i < NL; ++i) tn[i] = *reinterpret_cast<const float4*>(t1 + 4 * cl + 64 * i);
    }
    float* os = B.watt + ((size_t)n * T + t) * HWp;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float ss = 0.f, dot = 0.f;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const float p0 = m[u][i].x * t4[i].x, p1 = m[u][i].y * t4[i].y,
                    p2 = m[u][i].z * t4[i].z, p3 = m[u][i].w * t4[i].w;
        ss += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
        dot += p0 * e4[i].x + p1 * e4[i].y + p2 * e4[i].z + p3 * e4[i].w;
      }
      const float s2 = row16_sum(ss), d2 = row16_sum(dot);
      const int r = r0 + 4 * u + rg;
      if (cl == 0 && r < HW) os[r] = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;   // l2_normalize eps (A.4)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Deferred attention pooling (throughput mode): f_i = sum_hw a_i[hw] * feat[n, hw, :] for the
// questions whose root is Describe / SameProperty (nmn3_modules.py:438-441,485-486) -- THE HBM-bound
// kernel of the attention-module path: the [H*W, D] feature map (307 KB at CLEVR dims) is read once
// per question (both inputs of SameProperty from the same read) and nothing else moves.
// One workgroup per (channel part, question): POOLP parts of D / POOLP channels, so a launch with J
// jobs puts POOLP * J workgroups on the chip (a batch of 64 questions alone would be ~25 jobs; the
// super-bucket brings it to hundreds).  All feature loads of a thread are issued before the soft-max
// weights are read.  Questions without such a root exit after one 4-byte read.
// ---------------------------------------------------------------------------------------------
constexpr int POOLP = WALK_POOL_PARTS;
__global__ __launch_bounds__(256) void walk_pool_kernel(WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int part = blockIdx.x, q = blockIdx.y;
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int op = B.pjob[n];
  if (op == 0) return;
  const int HW = a.H * a.W, D = a.D, HWp = a.HWp;
  const int Dp = D / POOLP, ncol = Dp / 4, nrow = 256 / ncol;
  const int tid = threadIdx.x, lc = tid % ncol, lr = tid / ncol;
  const int nin = op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
  constexpr int PR = WALK_POOLK_ROWS;
  const float* fp = B.feat + (size_t)n * HW * D + part * Dp + 4 * lc;
  const int myrows = lr < nrow ? (HW - lr + nrow - 1) / nrow : 0;
  float4 fr[PR];
  {
    const unsigned rowstep = (unsigned)(nrow * D);
    unsigned off = (unsigned)(lr * D);
    const unsigned last = (unsigned)((myrows > 0 ? lr + (myrows - 1) * nrow : 0) * D);
#pragma unroll
    for (int qq = 0; qq < PR; ++qq) {
      fr[qq] = *reinterpret_cast<const float4*>(fp + min(off, last));
      off += rowstep;
    }
  }
  float* a0 = smem;                      // [HWp]
  float* a1 = a0 + HWp;                  // [HWp]
  float* stage = a1 + HWp;               // [nrow][2][Dp]
  const float* pw = B.pw + (size_t)n * 2 * HWp;
  for (int r = tid; r < HW; r += 256) {
    a0[r] = pw[r];
    a1[r] = nin == 2 ? pw[HWp + r] : 0.f;
  }
  __syncthreads();
  float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
  if (lr < nrow) {
#pragma unroll
    for (int qq = 0; qq < PR; ++qq) {
      if (qq < myrows) {
        const int r = lr + qq * nrow;
        const float w0 = a0[r], w1 = a1[r];
        acc0.x += w0 * fr[qq].x; acc0.y += w0 * fr[qq].y; acc0.z += w0 * fr[qq].z; acc0.w += w0 * fr[qq].w;
        acc1.x += w1 * fr[qq].x; acc1.y += w1 * fr[qq].y; acc1.z += w1 * fr[qq].z; acc1.w += w1 * fr[qq].w;
      }
    }
    *reinterpret_cast<float4*>(stage + (size_t)(lr * 2 + 0) * Dp + 4 * lc) = acc0;
    *reinterpret_cast<float4*>(stage + (size_t)(lr * 2 + 1) * Dp + 4 * lc) = acc1;
  }
  __syncthreads();
  float* out = B.pooled + (size_t)n * 2 * D + part * Dp;
  for (int i = tid; i < nin * Dp; i += 256) {
    const int which = i / Dp, c = i - which * Dp;
    float sacc = 0.f;
    for (int qq = 0; qq < nrow; ++qq) sacc += stage[(size_t)(qq * 2 + which) * Dp + c];
    out[(size_t)which * D + c] = sacc;
  }
}

// fc_att of the deferred pooling jobs (nmn3_modules.py:442-446, 487-490), grouped: every job of one
// weight set multiplies its pooled feature vector with the same [D, Mp] matrix (512 KB at CLEVR
// dimensions).  One workgroup per job pulled that matrix from L2 once per job (309 jobs: 158 MB, 34 us
// per 1024 questions); here a workgroup takes HG jobs of ONE weight set -- from the per-operator job lists
// that walk_light_kernel / walk_kernel append to when they hand a root over (WalkArgs::plist; scanning
// the launch's job codes per workgroup cost more than the products) -- and 64 of the Mp columns, so every weight row it
// fetches meets HG vectors, a workgroup's share of the matrix is 128 KB, and all of it is in flight at
// once: 32 k-groups x 16 column lanes, D / 32 rows per thread.
// Weight sets: y = 0 Describe (fc_att of input 0), 1 / 2 SameProperty input 0 / input 1.
// Rows go to pfc[n][input][Mp].
constexpr int HG = 4, FCW = 64, FKG = WT / (FCW / 4);          // jobs, columns, k-groups per workgroup
constexpr int FC_GROUPS = 96;                                  // job groups of the persistent grid
__global__ __launch_bounds__(WT) void walk_fcatt_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int y = blockIdx.y, c0 = blockIdx.z * FCW;
  const int ty = y == 0 ? 0 : 1;                  // list: Describe / SameProperty
  const int which = y == 2 ? 1 : 0;
  const int njob = min(a.cnt[3 + ty], a.pcap);    // (uniform: scalar load)
  if (blockIdx.x == 0 && y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    // the NEXT pass starts from empty lists (this pass's set stays readable for replays)
    for (int i = 0; i < WALK_CNT; ++i) a.cnt_next[i] = 0;
    if (a.hint) *a.hint = a.cnt[5];                // deepest nesting of this pass, for the host (no sync)
  }
  const int tid = threadIdx.x;
  const int D = a.D, Mp = a.Mp;
  float* x = smem;                                // [HG][D]
  float* part = x + (size_t)HG * D;               // [FKG][HG][FCW]
  const int wi = y == 0 ? 3 : y;                  // ModuleWeights::Watt: FSP, SameProperty 0 / 1, Describe
  const float* Wp = w.Watt[wi];
  const float* bm = w.batt[wi];
  const int cl = tid & (FCW / 4 - 1), kg = tid / (FCW / 4);
  const int kper = (D + FKG - 1) / FKG;
  const int k0 = kg * kper, k1 = min(D, k0 + kper);
  const unsigned col = (unsigned)(c0 + 4 * cl);   // (Mp % 64 == 0: every column of the part exists)
  constexpr int KU = 16;
  const int32_t* list = a.plist + (size_t)ty * a.pcap;
  for (int g = blockIdx.x; HG * g < njob; g += gridDim.x) {
    const int cnt = min(HG, njob - HG * g);
    float4 w4[KU];
    auto fetch = [&](int kq) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const unsigned k = (unsigned)min(kq + u, max(k1 - 1, 0));
        w4[u] = *reinterpret_cast<const float4*>(Wp + (k * (unsigned)Mp + col));
      }
    };
    fetch(k0);                                    // the first weight rows are in flight ...
    for (int i = tid; i < HG * D; i += WT) {      // ... while the pooled vectors of the jobs arrive
      const int j = i / D, k = i - j * D;
      float v = 0.f;
      if (j < cnt) {
        const int q = list[HG * g + j], kb = q / a.N, n = q - kb * a.N;
        v = a.b[kb].pooled[((size_t)n * 2 + which) * D + k];
      }
      x[i] = v;
    }
    __syncthreads();
    float4 acc[HG];
#pragma unroll
    for (int j = 0; j < HG; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kq = k0; kq < k1; kq += KU) {
      if (kq != k0) fetch(kq);
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (kq + u < k1) {
#pragma unroll
          for (int j = 0; j < HG; ++j) {
            const float xv = x[j * D + kq + u];
            acc[j].x += xv * w4[u].x; acc[j].y += xv * w4[u].y; acc[j].z += xv * w4[u].z;
            acc[j].w += xv * w4[u].w;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < HG; ++j)
      *reinterpret_cast<float4*>(part + ((size_t)(kg * HG + j) * FCW) + 4 * cl) = acc[j];
    __syncthreads();
    for (int i = tid; i < HG * FCW; i += WT) {
      const int j = i / FCW, c = i - j * FCW;
      if (j < cnt) {
        float r = bm[c0 + c];
        for (int qq = 0; qq < FKG; ++qq) r += part[(size_t)(qq * HG + j) * FCW + c];
        const int q = list[HG * g + j], kb = q / a.N, n = q - kb * a.N;
        a.b[kb].pfc[((size_t)n * 2 + which) * Mp + c0 + c] = r;
      }
    }
    __syncthreads();                              // x / part are reused by the next group
  }
}

// l2-normalised product with the text map and fc_eltwise of the deferred questions
// (nmn3_modules.py:447-450, 491-493): one workgroup per question, the walker's in-line arithmetic, on
// the fc_att rows walk_fcatt_kernel left in pfc.
__global__ __launch_bounds__(WT) void walk_heads_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int q = blockIdx.x;
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int op = B.pjob[n];
  if (op == 0) return;
  const int M = a.M, Mp = a.Mp, C = a.C;
  const int tid = threadIdx.x;
  float* ev = smem;                      // [Mp]
  float* rs = ev + Mp;                   // [32]
  float* scr = rs + 32;                  // [WT]
  const bool same = op == N2NMN_OP_SAME_PROPERTY;
  const float* am0 = B.pfc + (size_t)n * 2 * Mp;
  const float* am1 = am0 + Mp;
  const float* tm = B.ptm + (size_t)n * Mp;
  float lss = 0.f;
  for (int c = tid; c < Mp; c += WT) {
    float v = 0.f;
    if (c < M) {
      v = am0[c] * tm[c];
      if (same) v *= am1[c];
    }
    ev[c] = v;
    lss += v * v;
  }
  const float ss = wg_reduce<0>(lss, rs);
  const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  for (int c = tid; c < Mp; c += WT) ev[c] *= inv;
  __syncthreads();
  const int wi = same ? 5 : 6;
  fc_out(tid, ev, M, w.Wans[wi], w.bans[wi], C, B.scores + (size_t)n * C, scr);
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------
// Staged walker, passes of many questions (WalkArgs::staged).  What is left of a question's tree once
// the text maps and the Find / Filter epilogues run chip-wide falls into two classes:
//   heavy  Transform (2.3 MFLOP of fp32 MFMA) and FindSameProperty (a 307 KB feature pool, a 512 KB
//          fc_att stream and a 154 KB map epilogue): microseconds each on one CU;
//   light  And / Or / Filter's min / Scene, the answer operators (a few hundred flops on 150-float
//          maps) and the soft-max of the deferred pooling roots.
// In the one-workgroup-per-question walker the heavy nodes sat on each question's chain with one
// 235-VGPR workgroup per CU (four rounds of 256 questions per 1024: 100 us, 7 % of HBM).  Here
//   walk_heavy_kernel  runs every heavy node as a job of its own, one launch per NESTING LEVEL (level =
//                      heavy nodes below the node in its own subtree; lists per level and operator built
//                      by walk_tmap_kernel's plan step): the workgroup evaluates the node's input subtree
//                      from the Find / Filter logits and the lower levels' maps in `watt`, runs the
//                      operator with the walker's own device code, and writes the map to watt[n][t];
//   walk_light_kernel  one workgroup per question: every remaining node from `watt`, the answer
//                      logits or the deferred-pooling tables;
//   walk_kernel        only for the questions nested deeper than the pass launches levels for (fblist;
//                      the level count follows the previous passes: n2nmn_walk_layouts).
// Same arithmetic, operator by operator, as walk_kernel (shared device functions / copied statement
// by statement); the answer heads reduce in another order (logits within 2e-6 between the paths).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_prog(int tid, const WalkProg* src, WalkProg& P) {
  const int4* s4 = reinterpret_cast<const int4*>(src);
  int4* d4 = reinterpret_cast<int4*>(&P);
  if (tid < (int)(sizeof(WalkProg) / 16)) d4[tid] = s4[tid];
}


// the light operators of nodes [lo, hi] in token order, on maps that are already in the LDS arena (Find,
// Filter's find_result, and the heavy nodes of lower levels): eval_light_range without its loads
template <int NT>
__device__ __forceinline__ void eval_light_ops(int tid, const WalkProg& P, int lo, int hi, float* arena,
                                               int HW, int HWp) {
  for (int t = lo; t <= hi; ++t) {
    const int o = P.op[t] & 0x7f;
    float* outp = arena + (size_t)t * HWp;
    const float* in0 = P.in0[t] >= 0 ? arena + (size_t)P.in0[t] * HWp : nullptr;
    const float* in1 = P.in1[t] >= 0 ? arena + (size_t)P.in1[t] * HWp : nullptr;
    bool wrote = true;
    switch (o) {
      case N2NMN_OP_SCENE:                                       // :60-72
        for (int r = tid; r < HW; r += NT) outp[r] = 3.0f;
        break;
      case N2NMN_OP_AND:                                         // :218-236
        for (int r = tid; r < HW; r += NT) outp[r] = fminf(in0[r], in1[r]);
        break;
      case N2NMN_OP_OR:                                          // :238-256
        for (int r = tid; r < HW; r += NT) outp[r] = fmaxf(in0[r], in1[r]);
        break;
      case N2NMN_OP_FILTER:                                      // And(input_0, find_result) :129-130
        for (int r = tid; r < HW; r += NT) outp[r] = fminf(in0[r], outp[r]);
        break;
      default: wrote = false; break;
    }
    if (wrote) __syncthreads();                                  // (uniform: o is the same for all)
  }
}

// FindSameProperty (nmn3_modules.py:134-183) as chip-wide stages (round 5).  One workgroup per node ran
// its three dependent streams -- 307 KB feature pool, 512 KB fc_att weights, 154 KB map epilogue -- at ONE
// CU's pace (17-46 GB/s: 31.6 us per node, 104 of them per 1024 questions set the length of the launch).
// Now a node is WALK_POOL_PARTS stage-A items in the level's walk_heavy_kernel launch (next to the level's
// Transform halves: both need only maps of lower levels, and the stream-bound pooling parts and the
// matrix-core-bound Transform halves share the CUs) and WALK_FIND_PARTS stage-B items in the
// walk_fspepi_kernel launch behind it:
//   A (fsp_pool_rest): channel part `part` of the soft-max pooling (:170-172) -- 38 KB of the feature map,
//     every load in flight before the plan of the question has even arrived -- and, because fc_att is
//     linear (:173-176), that part's share of it: pooled[part] . W_att[rows of the part] -> fpart[n][t][part]
//     (64 KB of weights per item; the full product is the bias plus the eight shares, summed in part order
//     by stage B: no atomics, a fixed order);
//   B (fsp_epi_item): att = l2norm_c(M_fsp[r, c] * tmap[c] * am[c]) . w_e + b_e (:178-180) over the
//     operator's own conv_image map, 8 row parts per node streaming their rows like walk_find_kernel.
constexpr int HT = 256, HWV = HT / 64;                  // walk_heavy_kernel: 4 waves per work item
constexpr int FSP_PR = WALK_POOLK_ROWS;                 // feature rows of a thread (walk_pool_supported's bound)
constexpr int FSP_KU = 8;                               // fc_att weight rows in flight per thread
struct FspLoads { float4 fr[FSP_PR]; float4 w4[FSP_KU]; };

__device__ __forceinline__ void fsp_fetch_w(FspLoads& L, const float* Wp, int kb, int k0, int k1, int Mp,
                                            unsigned col) {
#pragma unroll
  for (int u = 0; u < FSP_KU; ++u) {
    const unsigned k = (unsigned)min(kb + u, max(k1 - 1, k0));
    L.w4[u] = *reinterpret_cast<const float4*>(Wp + (k * (unsigned)Mp + col));
  }
}

// the loads of a stage-A item that depend on nothing but (question, node, part)
__device__ __forceinline__ void fsp_pool_loads(int tid, const ModuleWeights& w, const WalkArgs& a,
                                               const WalkBatch& B, int n, int part, FspLoads& L) {
  const int HW = a.H * a.W, D = a.D, Mp = a.Mp;
  const int Dp = D / POOLP, ncol = Dp / 4, nrow = HT / ncol;
  const int lane = tid & 63, wid = tid >> 6;
  const int lc = tid % ncol, lr = tid / ncol;
  const float* fp = B.feat + (size_t)n * HW * D + part * Dp + 4 * lc;
  const int myrows = lr < nrow ? (HW - lr + nrow - 1) / nrow : 0;
  const unsigned rowstep = (unsigned)(nrow * D);
  unsigned off = (unsigned)(lr * D);
  const unsigned last = (unsigned)((myrows > 0 ? lr + (myrows - 1) * nrow : 0) * D);
#pragma unroll
  for (int qq = 0; qq < FSP_PR; ++qq) {
    L.fr[qq] = *reinterpret_cast<const float4*>(fp + min(off, last));
    off += rowstep;
  }
  const int kper = (Dp + HWV - 1) / HWV;
  const int k0 = part * Dp + wid * kper, k1 = min(part * Dp + Dp, k0 + kper);
  fsp_fetch_w(L, w.Watt[0], k0, k0, k1, Mp, (unsigned)min(4 * lane, Mp - 4));
}

// the rest of the item, once the input map of the node is in the arena
__device__ __forceinline__ void fsp_pool_rest(int tid, const ModuleWeights& w, const WalkArgs& a,
                                              const WalkBatch& B, int n, int t, int part, const float* in0,
                                              FspLoads& L, float* pp, float* sa0, float* scr) {
  const int HW = a.H * a.W, D = a.D, Mp = a.Mp, T = a.T;
  const int Dp = D / POOLP, ncol = Dp / 4, nrow = HT / ncol;
  const int lane = tid & 63, wid = tid >> 6;
  const int lc = tid % ncol, lr = tid / ncol;
  const int myrows = lr < nrow ? (HW - lr + nrow - 1) / nrow : 0;
  // soft-max over the H*W logits (:170-172): ONE wave, in registers (walk_light_kernel's form; H*W <= 192)
  if (wid == 0) {
    float v[3], lm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int r = lane + 64 * u;
      v[u] = r < HW ? in0[r] : -INFINITY;
      lm = fmaxf(lm, v[u]);
    }
    const float mx = wave_max(lm);
    float ls = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) { v[u] = lane + 64 * u < HW ? expf(v[u] - mx) : 0.f; ls += v[u]; }
    const float sum = wave_sum(ls);
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (lane + 64 * u < HW) sa0[lane + 64 * u] = v[u] / sum;
  }
  __syncthreads();
  float* stage = scr;                                    // [nrow][Dp]
  if (lr < nrow) {
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int qq = 0; qq < FSP_PR; ++qq) {
      if (qq < myrows) {
        const float w0 = sa0[lr + qq * nrow];
        acc0.x += w0 * L.fr[qq].x; acc0.y += w0 * L.fr[qq].y; acc0.z += w0 * L.fr[qq].z; acc0.w += w0 * L.fr[qq].w;
      }
    }
    *reinterpret_cast<float4*>(stage + (size_t)lr * Dp + 4 * lc) = acc0;
  }
  __syncthreads();
  for (int c = tid; c < Dp; c += HT) {
    float sacc = 0.f;
    for (int qq = 0; qq < nrow; ++qq) sacc += stage[(size_t)qq * Dp + c];
    pp[c] = sacc;
  }
  __syncthreads();
  // ---- this part's share of fc_att (:173-176): fpart[n][t][part][c] = sum_{k in part} pooled[k] W[k][c];
  // wave = k group, float4 columns over the lanes (the first FSP_KU weight rows were requested with the features)
  const int kper = (Dp + HWV - 1) / HWV;
  const int k0 = part * Dp + wid * kper, k1 = min(part * Dp + Dp, k0 + kper);
  float* red = scr;                                      // [HWV][256]
  float* dst = B.fpart + (((size_t)n * T + t) * POOLP + part) * Mp;
  for (int cb = 0; cb < Mp; cb += 256) {
    const unsigned col = (unsigned)min(cb + 4 * lane, Mp - 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = k0; kb < k1; kb += FSP_KU) {
      if (kb != k0 || cb != 0) fsp_fetch_w(L, w.Watt[0], kb, k0, k1, Mp, col);
#pragma unroll
      for (int u = 0; u < FSP_KU; ++u) {
        if (kb + u < k1) {
          const float xv = pp[kb + u - part * Dp];
          acc.x += xv * L.w4[u].x; acc.y += xv * L.w4[u].y; acc.z += xv * L.w4[u].z; acc.w += xv * L.w4[u].w;
        }
      }
    }
    *reinterpret_cast<float4*>(red + (size_t)wid * 256 + 4 * lane) = acc;
    __syncthreads();
    if (cb + tid < Mp) {
      float r = 0.f;
#pragma unroll
      for (int q = 0; q < HWV; ++q) r += red[q * 256 + tid];
      dst[cb + tid] = r;
    }
    __syncthreads();
  }
}

// Items of a level, 4 waves each: two per Transform node (pixel halves: a node's 2.3 MFLOP of fp32 MFMA are
// 9 k clocks of one CU's matrix pipe, and 300 nodes on 256 CUs leave a fifth of the CUs with two of them;
// 600 half nodes spread), then WALK_POOL_PARTS stage-A items per FindSameProperty node.  <= 128 VGPRs and
// 20 KB of LDS: four items per CU, so one item's dependent chain (list entry -> plan -> operators) runs
// under the others' MFMAs / feature streams.  Everything an item reads that does not depend on the plan
// -- ALL attention rows of the question (12.8 KB; the plan only says which are used), the text map, the
// features and fc_att weights -- is requested before the plan has arrived.
// (launch bounds: 4 waves per SIMD; one instantiation per (kernel size, pixel tiles per wave) so that no
// variant pays for another's registers)
struct HeavyLds { float *arena, *tml, *twl, *pp, *sa0, *scr; };
constexpr int HEAVY_NQ = (WALK_MAX_T * 192 / 4 + HT - 1) / HT;   // float4 of the arena per thread (T * HWp <= 32 * 192)

// the attention rows in FRONT of a node (its subtree is a token range that ends right before it), requested
// before the question's plan is known -- the plan only says which of them are used; kept in registers
// across the barrier that guards the plan's LDS copy
struct HeavyRows { float4 v0, v1, v2, v3, v4, v5; };
static_assert(HEAVY_NQ == 6, "HeavyRows holds HEAVY_NQ float4");
__device__ __forceinline__ HeavyRows heavy_rows_load(int tid, const float* watt_q, int nq4) {
  const float4* wq4 = reinterpret_cast<const float4*>(watt_q);
  HeavyRows r;
  r.v0 = wq4[min(tid, nq4 - 1)];          r.v1 = wq4[min(tid + HT, nq4 - 1)];
  r.v2 = wq4[min(tid + 2 * HT, nq4 - 1)]; r.v3 = wq4[min(tid + 3 * HT, nq4 - 1)];
  r.v4 = wq4[min(tid + 4 * HT, nq4 - 1)]; r.v5 = wq4[min(tid + 5 * HT, nq4 - 1)];
  return r;
}
__device__ __forceinline__ void heavy_rows_store(int tid, float* arena, int nq4, const HeavyRows& r) {
  float4* a4 = reinterpret_cast<float4*>(arena);
  if (tid < nq4) a4[tid] = r.v0;
  if (tid + HT < nq4) a4[tid + HT] = r.v1;
  if (tid + 2 * HT < nq4) a4[tid + 2 * HT] = r.v2;
  if (tid + 3 * HT < nq4) a4[tid + 3 * HT] = r.v3;
  if (tid + 4 * HT < nq4) a4[tid + 4 * HT] = r.v4;
  if (tid + 5 * HT < nq4) a4[tid + 5 * HT] = r.v5;
}
// is node t of the plan a node of this operator at this level (a stale or foreign list entry is skipped)?
__device__ __forceinline__ bool heavy_node_ok(const WalkProg& P, int t, int op, int lv) {
  return P.valid && !P.fallback && t < P.nn && (P.op[t] & 0x7f) == op && P.hd[t] == lv + 1 && P.in0[t] >= 0;
}

// pixel half `half` of Transform node (n, t)
template <int KS, int PTW>
__device__ __forceinline__ void heavy_transform_item(int tid, const ModuleWeights& w, const WalkArgs& a,
                                                     const WalkBatch& B, WalkProg& P, int q, int n, int t,
                                                     int half, int lv, const HeavyLds& S) {
  const int HW = a.H * a.W, Mp = a.Mp, HWp = a.HWp, T = a.T;
  const int nq4 = t * HWp / 4;                          // rows [0, t)  (HWp % 4 == 0; t >= 1: the node has an input)
  const HeavyRows wq = heavy_rows_load(tid, B.watt + (size_t)n * T * HWp, nq4);
  float4 tmv = make_float4(0.f, 0.f, 0.f, 0.f), wev = tmv;
  if (4 * tid < Mp) {                                   // (Mp <= 4 * HT = 1024: MAXCI)
    tmv = *reinterpret_cast<const float4*>(B.tmap + ((size_t)t * a.N + n) * Mp + 4 * tid);
    wev = *reinterpret_cast<const float4*>(w.we[2] + 4 * tid);
  }
  __syncthreads();                                     // the previous item's readers of P / LDS are done
  load_prog(tid, B.prog + n, P);
  heavy_rows_store(tid, S.arena, nq4, wq);
  if (4 * tid < Mp) {
    *reinterpret_cast<float4*>(S.tml + 4 * tid) = tmv;
    *reinterpret_cast<float4*>(S.twl + 4 * tid) =
        make_float4(tmv.x * wev.x, tmv.y * wev.y, tmv.z * wev.z, tmv.w * wev.w);
  }
  __syncthreads();
  if (!heavy_node_ok(P, t, N2NMN_OP_TRANSFORM, lv)) return;
  const int i0 = P.in0[t];
  // the node's input map: its light subtree (level >= 1: the Transform / FindSameProperty maps of the lower
  // levels inside the subtree came from watt with the rest)
  eval_light_ops<HT>(tid, P, P.lo[i0], i0, S.arena, HW, HWp);
  // debug timeline (n2nmn_debug_walk_timeline), first half only: [0] operands in LDS, [1] padded map built,
  // [2] matrix phase done, [3] map written
  long long* tl = (a.timeline && half == 0) ? a.timeline + ((size_t)q * MAXT + t) * 4 : nullptr;
  if (tl && threadIdx.x == 0) tl[0] = clock64();
  float* outp = S.arena + (size_t)t * HWp;
  walk_transform_t<KS, PTW, HWV>(tid, w, a, S.arena + (size_t)i0 * HWp, S.tml, S.twl, outp, S.scr, tl, half);   // :185-216
  float* dst = B.watt + ((size_t)n * T + t) * HWp;
  for (int pl = tid; pl < PTW * 16; pl += HT) {
    const int pxl = half * PTW * 16 + pl;
    if (pxl < HW) dst[pxl] = outp[pxl];
  }
  if (tl && threadIdx.x == 0) tl[3] = clock64();
}

// Stage B of FindSameProperty node (n, t), row part `part`: the Find-type epilogue over the operator's own
// conv_image map, with tmap (.) (b_att + the eight fc_att shares of stage A) in the place of Find's text map
template <int CI>
__device__ __forceinline__ void fsp_epi_item(int tid, const ModuleWeights& w, const WalkArgs& a,
                                             const WalkBatch& B, int n, int t, int part, int lv) {
  const int lane = tid & 63;
  const int T = a.T, HW = a.H * a.W, Mp = a.Mp, HWp = a.HWp;
  const int rpp = (HW + WALK_FIND_PARTS - 1) / WALK_FIND_PARTS;
  const int r0 = part * rpp, r1 = min(HW, r0 + rpp);
  if (r0 >= r1) return;
  const float* Mbuf = B.mfsp + (size_t)n * HW * Mp;
  const FindRows<CI> pre = find_load<CI>(Mbuf, r0 + (tid >> 6), r1, lane, Mp);
  const WalkProg* P = B.prog + n;                      // uniform address: scalar loads
  if (!P->valid || P->fallback || t >= P->nn || (P->op[t] & 0x7f) != N2NMN_OP_FIND_SAME_PROPERTY ||
      P->hd[t] != lv + 1)
    return;
  float4 t4[1][CI], e4[CI];
  const float* tm = B.tmap + ((size_t)t * a.N + n) * Mp;
  const float* fp = B.fpart + ((size_t)n * T + t) * POOLP * Mp;
#pragma unroll
  for (int i = 0; i < CI; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < Mp) {
      float4 am = *reinterpret_cast<const float4*>(w.batt[0] + c);
      float4 sh[POOLP];
#pragma unroll
      for (int p = 0; p < POOLP; ++p) sh[p] = *reinterpret_cast<const float4*>(fp + (size_t)p * Mp + c);
#pragma unroll
      for (int p = 0; p < POOLP; ++p) { am.x += sh[p].x; am.y += sh[p].y; am.z += sh[p].z; am.w += sh[p].w; }
      const float4 tv = *reinterpret_cast<const float4*>(tm + c);
      t4[0][i] = make_float4(tv.x * am.x, tv.y * am.y, tv.z * am.z, tv.w * am.w);
      e4[i] = *reinterpret_cast<const float4*>(w.we[1] + c);
    } else {
      t4[0][i] = make_float4(0.f, 0.f, 0.f, 0.f); e4[i] = t4[0][i];
    }
  }
  float* os[1] = {B.watt + ((size_t)n * T + t) * HWp};
  if (rpp <= FindUnroll<CI>::value * FW) find_rows_core<CI, 1, true>(tid, w.be[1][0], e4, t4, Mbuf, os, r0, r1, Mp, pre);
  else find_rows_core<CI, 1, false>(tid, w.be[1][0], e4, t4, Mbuf, os, r0, r1, Mp, pre);
}

// channel part `part` of stage A of FindSameProperty node (q, t)
__device__ __forceinline__ void fsp_pool_item(int tid, const ModuleWeights& w, const WalkArgs& a, WalkProg& P,
                                              int q, int t, int part, int lv, float* arena, float* pp,
                                              float* sa0, float* scr) {
  const int HW = a.H * a.W, HWp = a.HWp, T = a.T;
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  FspLoads FL;
  fsp_pool_loads(tid, w, a, B, n, part, FL);
  // rows [0, t): the node's subtree lies in front of it.  The first HT float4 (6 rows of 160) ride in
  // registers across the plan's barrier; a deeper subtree's remaining rows are fetched behind it
  const int nq4 = t * HWp / 4;
  const float4* wq4 = reinterpret_cast<const float4*>(B.watt + (size_t)n * T * HWp);
  const float4 wr0 = wq4[min(tid, nq4 - 1)];
  __syncthreads();                                     // the previous item's readers of P / LDS are done
  load_prog(tid, B.prog + n, P);
  {
    float4* a4 = reinterpret_cast<float4*>(arena);
    if (tid < nq4) a4[tid] = wr0;
    for (int i = tid + HT; i < nq4; i += HT) a4[i] = wq4[i];
  }
  __syncthreads();
  if (!heavy_node_ok(P, t, N2NMN_OP_FIND_SAME_PROPERTY, lv)) return;
  const int i0 = P.in0[t];
  // (level >= 1: the Transform / FindSameProperty maps of the lower levels inside the subtree came from watt)
  eval_light_ops<HT>(tid, P, P.lo[i0], i0, arena, HW, HWp);
  fsp_pool_rest(tid, w, a, B, n, t, part, arena + (size_t)i0 * HWp, FL, pp, sa0, scr);
}

// First launch of a nesting level: walk_heavy_kernel, a persistent grid (4 workgroups of 4 waves per CU) over
// WALK_POOL_PARTS stage-A items per FindSameProperty node of the level, then two items per Transform node
// (pixel halves: a node's 2.3 MFLOP of fp32 MFMA are 9 k clocks of one CU's matrix pipe; 600 half nodes
// spread where 300 nodes left a fifth of the CUs with two).  Everything in it needs only the Find / Filter
// logits and the maps of lower levels.  <= 128 VGPRs and 20 KB of LDS per item: one item's dependent chain
// (list entry -> plan -> operators) runs under the others' MFMAs / feature streams, and everything an item
// reads that does not depend on the plan -- the attention rows in front of the node, its text map, the
// features and fc_att weights of its part -- is requested before the plan has arrived.
// Order and split were measured (profiles/r05_notes.md): with the Transform halves in the SECOND launch next
// to stage B (round 5's first arrangement) the level cost 30.7 us of back-to-back replays, this way 28.8;
// Transform halves in front of the stage-A items 29.4.
// (launch bounds: 4 waves per SIMD; one instantiation per (kernel size, pixel tiles per wave) so that no
// variant pays for another's registers)
template <int KS, int PTW>
__global__ __launch_bounds__(HT, 4) void walk_heavy_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ __attribute__((aligned(16))) WalkProg P;
  const int T = a.T;
  HeavyLds S;
  S.arena = smem;                                      // [T][HWp]
  S.tml = S.arena + (size_t)T * a.HWp;                 // [Mp] text map of the node
  S.twl = S.tml + a.Mp;                                // [Mp] text map (.) w_e
  S.pp = S.twl + a.Mp;                                 // [D / POOL_PARTS] pooled features of the part
  S.sa0 = S.pp + a.D / POOLP;                          // [HWp]
  S.scr = S.sa0 + a.HWp;                               // operator scratch
  const int lv = a.hlevel;
  const int ctr = lv == 0 ? 0 : 6 + 2 * lv, cfs = lv == 0 ? 2 : 7 + 2 * lv;
  const int cap = (a.hoff[lv + 1] - a.hoff[lv]) / 2;
  const int nfsp = min(a.cnt[cfs], cap), ntr = min(a.cnt[ctr], cap);
  const int32_t* jtr = a.hjobs + a.hoff[lv];
  const int32_t* jfs = jtr + cap;
  const int nA = POOLP * nfsp, nT = 2 * ntr;
  for (int j = blockIdx.x; j < nA + nT; j += gridDim.x) {
    // (an opaque copy per item: the per-thread addresses of one item's 18 loads must not be hoisted out of
    // the loop, where they would stay live across everything else -- see walk_kernel)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const bool tr = j >= nA;
    const int jj = tr ? j - nA : j;
    const int job = tr ? jtr[jj >> 1] : jfs[jj / POOLP];
    const int part = tr ? (jj & 1) : jj % POOLP;
    const int q = job >> 8, t = job & 0xff;
    if (q < 0 || q >= a.K * a.N || t >= T || t < 1) continue;        // (a stale list entry)
    const int kb = q / a.N, n = q - kb * a.N;
    if (tr) heavy_transform_item<KS, PTW>(tid, w, a, a.b[kb], P, q, n, t, part, lv, S);
    else fsp_pool_item(tid, w, a, P, q, t, part, lv, S.arena, S.pp, S.sa0, S.scr);
  }
}

// Second launch of a nesting level: stage B of its FindSameProperty nodes, WALK_FIND_PARTS row parts per node
// streaming the operator's own conv_image map like walk_find_kernel (persistent grid; the eight fc_att
// shares of stage A came from the launch before).
template <int CI>
__global__ __launch_bounds__(HT, CI == 1 ? 4 : 2) void walk_fspepi_kernel(ModuleWeights w, WalkArgs a) {
  const int lv = a.hlevel;
  const int cfs = lv == 0 ? 2 : 7 + 2 * lv;
  const int cap = (a.hoff[lv + 1] - a.hoff[lv]) / 2;
  const int nfsp = min(a.cnt[cfs], cap);
  const int32_t* jfs = a.hjobs + a.hoff[lv] + cap;
  for (int j = blockIdx.x; j < WALK_FIND_PARTS * nfsp; j += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                        // (see walk_kernel)
    const int job = jfs[j / WALK_FIND_PARTS], part = j % WALK_FIND_PARTS;
    const int q = job >> 8, t = job & 0xff;
    if (q < 0 || q >= a.K * a.N || t >= a.T || t < 1) continue;      // (a stale list entry)
    const int kb = q / a.N, n = q - kb * a.N;
    fsp_epi_item<CI>(tid, w, a, a.b[kb], n, t, part, lv);
  }
}

// One workgroup per question: every node that is not a Transform / FindSameProperty, the answer operator
// or the hand-over of a pooling root.  The question's chain is a sequence of dependent round trips (plan ->
// maps -> answer weights); the last one leaves the chain: the first group of the answer fc's weights / the
// root's text map is requested as soon as the plan names the root operator, under the tree evaluation.
template <int NT>
__device__ __forceinline__ void light_item(int tid, const ModuleWeights& w, const WalkArgs& a, WalkProg& P,
                                           int q, float* smem) {
  const int kb = q / a.N, n = q - kb * a.N;
  const WalkBatch& B = a.b[kb];
  const int HW = a.H * a.W, Mp = a.Mp, HWp = a.HWp, T = a.T, C = a.C;
  float* arena = smem;                                 // [T][HWp]
  float* rs = arena + (size_t)T * HWp;                 // [32]
  float* scr = rs + 32;                                // answer features + fc_out partial sums
  // debug timeline: row MAXT - 1 of the question: [0] start, [1] tree evaluated, [3] end
  long long* tl = a.timeline ? a.timeline + ((size_t)q * MAXT + (MAXT - 1)) * 4 : nullptr;
  if (tl && tid == 0) tl[0] = clock64();
  load_prog(tid, B.prog + n, P);
  __syncthreads();
  if (P.valid && P.fallback) return;                   // walk_kernel serves this question (fblist)
  float* srow = B.scores + (size_t)n * C;
  if (tid == 0 && B.validity) B.validity[n] = P.valid;
  if (tid == 0) B.pjob[n] = 0;
  if (!P.valid) {                                      // INVALID_EXPR: zero logits (nmn3_model.py:146,155)
    for (int c = tid; c < C; c += NT) srow[c] = 0.f;
    return;
  }
  const int nn = P.nn;
  const int t = nn - 1;
  const int op = P.op[t] & 0x7f;
  const bool pool_root = op == N2NMN_OP_DESCRIBE || op == N2NMN_OP_SAME_PROPERTY;
  // ---- what the root needs from memory, requested now
  int F = 0, wi = 0;
  if (op == N2NMN_OP_EXIST) { F = 3; wi = 0; }
  else if (op == N2NMN_OP_COUNT) { F = HW + 2; wi = 1; }
  else if (!pool_root) { F = 2 * HW + 4; wi = op == N2NMN_OP_EQUAL_NUM ? 2 : (op == N2NMN_OP_MORE_NUM ? 3 : 4); }
  FcFirst pre{};
  float4 tm4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* tsrc = B.tmap + ((size_t)t * a.N + n) * Mp;
  if (pool_root) { if (4 * tid < Mp) tm4 = *reinterpret_cast<const float4*>(tsrc + 4 * tid); }
  else pre = fc_out_first<NT>(tid, F, w.Wans[wi], C);
  if (a.stats && tid == 0) {                           // the counters walk_kernel keeps (profiling only)
    unsigned long long cf = 0, cpi = 0, cp = 0, ct = 0, ctr = 0, nfind = 0;
    for (int tt = 0; tt < nn; ++tt) {
      const int o = P.op[tt] & 0x7f;
      const bool f = o == N2NMN_OP_FIND || o == N2NMN_OP_FILTER || o == N2NMN_OP_FIND_SAME_PROPERTY;
      const bool p = o == N2NMN_OP_FIND_SAME_PROPERTY || o == N2NMN_OP_SAME_PROPERTY || o == N2NMN_OP_DESCRIBE;
      cf += o == N2NMN_OP_FIND_SAME_PROPERTY; cp += p;
      cpi += p ? (o == N2NMN_OP_SAME_PROPERTY ? 2 : 1) : 0;
      ct += (f || p || o == N2NMN_OP_TRANSFORM); ctr += o == N2NMN_OP_TRANSFORM;
      nfind += o == N2NMN_OP_FIND || o == N2NMN_OP_FILTER;
    }
    atomicAdd(a.stats + 8, (nfind + 3) / 4);
    atomicAdd(a.stats + 0, cf); atomicAdd(a.stats + 1, cpi); atomicAdd(a.stats + 2, cp);
    atomicAdd(a.stats + 3, ct); atomicAdd(a.stats + 4, ctr); atomicAdd(a.stats + 5, 1ull);
    if (pool_root) {
      atomicAdd(a.stats + 6, 1ull); atomicAdd(a.stats + 7, op == N2NMN_OP_SAME_PROPERTY ? 2ull : 1ull);
    }
  }
  // every attention node of the tree (the root is the only answer node of a valid layout): the rows that
  // exist in `watt` -- Find / Filter logits (walk_find), Transform / FindSameProperty maps (walk_heavy / walk_fspepi) --
  // with every load in flight, then the light operators in token order.  (Requesting ALL T rows before the
  // plan arrives, as the heavy items do for the rows in front of their node, was measured slower here: 12.8
  // KB per question where the template mix needs 0.6 - 1.8.)
  {
    const float* watt_q = B.watt + (size_t)n * T * HWp;
    for (int i = tid; i < (nn - 1) * HWp; i += NT) {
      const int tt = i / HWp, r = i - tt * HWp;
      const int o = P.op[tt] & 0x7f;
      const bool mat = o == N2NMN_OP_FIND || o == N2NMN_OP_FILTER || o == N2NMN_OP_TRANSFORM ||
                       o == N2NMN_OP_FIND_SAME_PROPERTY;
      if (mat && r < HW) arena[(size_t)tt * HWp + r] = watt_q[(size_t)tt * HWp + r];
    }
    __syncthreads();
  }
  eval_light_ops<NT>(tid, P, 0, nn - 2, arena, HW, HWp);
  if (tl && tid == 0) tl[1] = clock64();
  const float* in0 = P.in0[t] >= 0 ? arena + (size_t)P.in0[t] * HWp : nullptr;
  const float* in1 = P.in1[t] >= 0 ? arena + (size_t)P.in1[t] * HWp : nullptr;
  const int lane = tid & 63, wid = tid >> 6;
  if (pool_root) {
    // deferred pooling root: soft-max weights, text map and job code for walk_pool / walk_heads
    // (:432-437,482-484).  A map has H*W = 150 values: ONE wave reduces it in registers (three values
    // per lane, DPP wave reductions) -- the block-wide reductions of walk_kernel cost two barriers and
    // an LDS round trip each, 6 k clocks of a 11 k clock question (tools/staged_timeline.py)
    const int nin = op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
    float* pw = B.pw + (size_t)n * 2 * HWp;
    if (wid < nin) {                                   // wave 0: input 0, wave 1: input 1
      const float* src = wid == 0 ? in0 : in1;
      float v[3], lm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int r = lane + 64 * u;
        v[u] = r < HW ? src[r] : -INFINITY;
        lm = fmaxf(lm, v[u]);
      }
      const float mx = wave_max(lm);
      float ls = 0.f;
#pragma unroll
      for (int u = 0; u < 3; ++u) { v[u] = lane + 64 * u < HW ? expf(v[u] - mx) : 0.f; ls += v[u]; }
      const float sum = wave_sum(ls);
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (lane + 64 * u < HW) pw[wid * HWp + lane + 64 * u] = v[u] / sum;
    }
    float* ptm = B.ptm + (size_t)n * Mp;
    if (4 * tid < Mp) *reinterpret_cast<float4*>(ptm + 4 * tid) = tm4;
    for (int c = 4 * (tid + NT); c < Mp; c += 4 * NT)      // (Mp > 4 * NT: not at CLEVR dimensions)
      *reinterpret_cast<float4*>(ptm + c) = *reinterpret_cast<const float4*>(tsrc + c);
    if (tid == 0) {
      B.pjob[n] = op;
      if (a.plist) {                                   // the job lists walk_fcatt_kernel works from
        const int ty = op == N2NMN_OP_SAME_PROPERTY ? 1 : 0;
        const int j = atomicAdd(a.cnt + 3 + ty, 1);
        if (j < a.pcap) a.plist[ty * a.pcap + j] = q;
      }
    }
    if (tl && tid == 0) tl[3] = clock64();
    return;
  }
  // Exist (:258-280), Count (:282-304), EqualNum / MoreNum / LessNum (:306-400)
  float* x = scr;                                  // up to 2*HW + 4 features
  float* red = x + ((2 * HW + 4 + 3) & ~3);
  const int nin = (op == N2NMN_OP_EXIST || op == N2NMN_OP_COUNT) ? 1 : 2;
  // min / max / sum of each input by ONE wave (see above); it also writes the features
  if (wid < nin) {
    const float* src = wid == 0 ? in0 : in1;
    float lmn = INFINITY, lmx = -INFINITY, lsm = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int r = lane + 64 * u;
      if (r < HW) {
        const float v = src[r];
        x[wid * (HW + 2) + r] = v;                   // row-major y*W + x flatten (:297)
        lmn = fminf(lmn, v); lmx = fmaxf(lmx, v); lsm += v;
      }
    }
    const float mn = wave_min(lmn), mx = wave_max(lmx), sm = wave_sum(lsm);
    if (lane == 0) { rs[4 * wid] = mn; rs[4 * wid + 1] = mx; rs[4 * wid + 2] = sm; }
  }
  __syncthreads();
  if (op == N2NMN_OP_EXIST) {
    if (tid == 0) { const float mn = rs[0], mx = rs[1], sm = rs[2]; x[0] = mn; x[1] = sm / (float)HW; x[2] = mx; }
  } else if (op == N2NMN_OP_COUNT) {
    if (tid == 0) { x[HW] = rs[0]; x[HW + 1] = rs[1]; }
  } else {
    if (tid == 0) {
      x[HW] = rs[0]; x[HW + 1] = rs[1];
      x[2 * HW + 2] = rs[4]; x[2 * HW + 3] = rs[5];
    }
  }
  __syncthreads();
  if (tl && tid == 0) tl[2] = clock64();
  fc_out_t<true, NT>(tid, x, F, w.Wans[wi], w.bans[wi], C, srow, red, pre);
  if (tl && tid == 0) tl[3] = clock64();
}


__global__ __launch_bounds__(HT) void walk_light_kernel(ModuleWeights w, WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ __attribute__((aligned(16))) WalkProg P;
  light_item<HT>(threadIdx.x, w, a, P, blockIdx.x, smem);
}

}  // namespace

// First launch of level a.hlevel: FindSameProperty stage A + Transform halves
void launch_walk_heavy(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const int cap = (a.hoff[a.hlevel + 1] - a.hoff[a.hlevel]) / 2;
  const int pad = a.ksize / 2;
  // LDS (floats): arena + text map (twice) + pooled part + soft-max row + the larger of Transform's padded
  // map / fold buffer and the pooling scratch
  const size_t tr = (((size_t)(a.H + 2 * pad) * (a.W + 2 * pad) + 4) & ~3) + (size_t)TR_CG * 96 * 2;
  const size_t smem = sizeof(float) * ((size_t)a.T * a.HWp + 2 * (size_t)a.Mp + a.D / POOLP + a.HWp +
                                       std::max<size_t>(tr, (size_t)4 * HT) + 64);
  // a persistent grid over the level's items: four workgroups per CU's worth of ids, each takes items
  // id, id + grid, ... (the list lengths live on the device)
  const int grid = (int)std::min<long>(1024, std::max<long>(1, (long)cap * (POOLP + 2)));
  auto go = [&](auto kern, std::atomic<uint64_t>& done) {
    if (smem > 64 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem, done);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(HT), smem, s, w, a);
  };
  static std::atomic<uint64_t> dn[4];
  const bool p5 = (a.H * a.W + 15) / 16 <= 10;     // pixel tiles per wave (walk_transform)
  if (a.ksize == 5) { if (p5) go(walk_heavy_kernel<5, 5>, dn[0]); else go(walk_heavy_kernel<5, 6>, dn[1]); }
  else { if (p5) go(walk_heavy_kernel<3, 5>, dn[2]); else go(walk_heavy_kernel<3, 6>, dn[3]); }
}

// Second launch of level a.hlevel: FindSameProperty stage B
void launch_walk_fspepi(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const int cap = (a.hoff[a.hlevel + 1] - a.hoff[a.hlevel]) / 2;
  const int grid = (int)std::min<long>(1024, std::max<long>(1, (long)cap * WALK_FIND_PARTS));   // (768: +1.2 us; 1536 / 2048: no change)
  if (a.Mp <= 256) hipLaunchKernelGGL(walk_fspepi_kernel<1>, dim3(grid), dim3(HT), 0, s, w, a);
  else hipLaunchKernelGGL(walk_fspepi_kernel<4>, dim3(grid), dim3(HT), 0, s, w, a);
}

void launch_walk_light(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const int HW = a.H * a.W;
  const size_t smem = sizeof(float) * ((size_t)a.T * a.HWp + 32 + (size_t)((2 * HW + 4 + 3) & ~3) + HT);
  hipLaunchKernelGGL(walk_light_kernel, dim3(a.K * a.N), dim3(HT), smem, s, w, a);
}

void launch_walk_pool(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const int Dp = a.D / POOLP, nrow = 256 / (Dp / 4);
  const size_t smem = sizeof(float) * (2 * (size_t)a.HWp + (size_t)nrow * 2 * Dp);
  hipLaunchKernelGGL(walk_pool_kernel, dim3(POOLP, a.K * a.N), dim3(256), smem, s, a);
  (void)w;
}

void launch_walk_heads(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  // fc_att of all deferred jobs, grouped by weight set, then the heads (one small workgroup each)
  const int QN = a.K * a.N;
  const size_t fsm = sizeof(float) * ((size_t)HG * a.D + (size_t)FKG * HG * FCW);
  static std::atomic<uint64_t> done{0};
  if (fsm > 64 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(walk_fcatt_kernel), (int)fsm, done);
  const int groups = std::min(FC_GROUPS, (QN + HG - 1) / HG);
  hipLaunchKernelGGL(walk_fcatt_kernel, dim3(groups, 3, a.Mp / FCW), dim3(WT), fsm, s, w, a);
  const size_t smem = sizeof(float) * ((size_t)a.Mp + 32 + WT);
  hipLaunchKernelGGL(walk_heads_kernel, dim3(QN), dim3(WT), smem, s, w, a);
}

int walk_pool_supported(int H, int W, int D) {
  if (D % (4 * POOLP) != 0) return 0;
  const int ncol = D / POOLP / 4;
  if (ncol > 256 || 256 % ncol != 0) return 0;
  const int nrow = 256 / ncol;
  return (H * W + nrow - 1) / nrow <= WALK_POOLK_ROWS;
}

int walk_supported(int H, int W, int D, int M, int Mp, int HWp, int E, int C, int T, int ksize,
                   int T_enc) {
  const int HW = H * W;
  if (D % 4 != 0 || D / 4 > WT || WT % (D / 4) != 0) return 0;
  const int nrow = WT / (D / 4);
  if ((HW + nrow - 1) / nrow > WALK_POOL_ROWS) return 0;
  if (Mp > 256 * MAXCI || Mp % 4 != 0 || C > WT || T > MAXT) return 0;
  if (ksize != 3 && ksize != 5) return 0;
  const int PG = (HW + 63) / 64;
  if (PG > WALK_MAX_PIXEL_GROUPS) return 0;
  const size_t bytes = sizeof(float) * walk_lds_floats(T, HWp, Mp, E, D, M, ksize, H, W, C, T_enc);
  return bytes <= 150 * 1024;
}

void launch_walk_textmap(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const int Ep = (a.E + 3) & ~3;
  const size_t smem = sizeof(float) * ((size_t)TMG * Ep + (size_t)WW * TMG * 256);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(walk_textmap_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(walk_textmap_kernel, dim3((a.N + TMG - 1) / TMG, 5, a.T * a.K), dim3(WT), smem,
                     s, w, a);
}

void launch_walk_tmap(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const int Tep = (a.T_enc + 3) & ~3;
  const size_t smem = sizeof(float) * ((size_t)Tep + (size_t)a.T * Tep);
  hipLaunchKernelGGL(walk_tmap_kernel, dim3(a.K * a.N), dim3(FT), smem, s, w, a);
}

void launch_walk_find(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  if (a.Mp <= 256 && a.Mp % 64 == 0) {                 // sixteen lanes per row (walk_find16_kernel)
    const dim3 g16(a.K * a.N, (a.H * a.W + F16_ROWS - 1) / F16_ROWS);
    switch (a.Mp / 64) {
      case 1: hipLaunchKernelGGL(walk_find16_kernel<1>, g16, dim3(FT), 0, s, w, a); break;
      case 2: hipLaunchKernelGGL(walk_find16_kernel<2>, g16, dim3(FT), 0, s, w, a); break;
      case 3: hipLaunchKernelGGL(walk_find16_kernel<3>, g16, dim3(FT), 0, s, w, a); break;
      default: hipLaunchKernelGGL(walk_find16_kernel<4>, g16, dim3(FT), 0, s, w, a); break;
    }
    return;
  }
  const dim3 grid(a.K * a.N, WALK_FIND_PARTS);
  const int ci = (a.Mp + 255) / 256;
  const int rpp = (a.H * a.W + WALK_FIND_PARTS - 1) / WALK_FIND_PARTS;      // rows of a workgroup
  if (ci == 1) {
    if (rpp <= FindUnroll<1>::value * FW) hipLaunchKernelGGL((walk_find_kernel<1, true>), grid, dim3(FT), 0, s, w, a);
    else hipLaunchKernelGGL((walk_find_kernel<1, false>), grid, dim3(FT), 0, s, w, a);
  } else if (ci == 2) {
    hipLaunchKernelGGL((walk_find_kernel<2, false>), grid, dim3(FT), 0, s, w, a);
  } else {
    hipLaunchKernelGGL((walk_find_kernel<4, false>), grid, dim3(FT), 0, s, w, a);
  }
}

void launch_walk(const ModuleWeights& w, const WalkArgs& a, hipStream_t s) {
  const size_t smem = sizeof(float) * walk_lds_floats(a.T, a.HWp, a.Mp, a.E, a.D, a.M, a.ksize,
                                                      a.H, a.W, a.C, a.T_enc);
  auto go = [&](auto kern) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(a.staged ? std::min(a.K * a.N, 64) : a.K * a.N), dim3(WT), smem, s, w, a);
  };
  const int ci = (a.Mp + 255) / 256;
  if (ci == 1) go(walk_kernel<1>);
  else if (ci == 2) go(walk_kernel<2>);
  else go(walk_kernel<4>);
}

}  // namespace n2nmn
