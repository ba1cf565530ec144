// Module-operator kernels (models_clevr/nmn3_modules.py:60-495), batched per level stage by the
// scheduler in schedule.cpp instead of per (module type, depth) like TensorFlow-Fold.
//
//   textmap  : fc_text / text_fc of every node with a text parameter           (:101,161,209,424,479)
//   att_ops  : Scene, Find / Filter / FindSameProperty epilogues (l2-normalise + conv_eltwise on
//              the hoisted conv_image map), Transform, And, Or, Exist, Count, Equal/More/LessNum
//   pool     : spatial softmax + attention-weighted feature sum + partial fc_att
//              (FindSameProperty :170-176, SameProperty :432-446, Describe :482-490)
//   heads    : Describe / SameProperty  l2-normalise + fc_eltwise               (:448-450,492-493)
//
// Data layout: image features stay in the reference's NHWC [N, H*W, D] fp32 and are indexed in
// place by batch_idx (Fold's tf.gather copy of [Nb,H,W,D], nmn3_modules.py:49-51, never exists);
// attention maps live in an arena [node][HWp] in HBM/L2; text maps [tslot][Mp] and conv_image
// maps [image slot][HW][Mp] are zero padded to Mp = round_up(map_dim, 64) so float4 lanes need no
// tail handling.
#include <algorithm>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

constexpr int MT = 256;   // threads per workgroup for every module kernel

// out[j] = b[j] + sum_f x[f] * Wm[f*C + j]   (x in LDS).  256 threads = nsl slices x C columns.
// `partial` needs >= 256 floats of LDS.  All threads must call.
__device__ void fc_small(const float* x, int F, const float* __restrict__ Wm,
                         const float* __restrict__ b, int C, float* __restrict__ out,
                         float* partial) {
  const int tid = threadIdx.x;
  if (C <= MT) {
    const int nsl = MT / C;
    const int j = tid % C, sl = tid / C;
    float s0 = 0.f, s1 = 0.f;
    if (sl < nsl) {
      int f = sl;
#pragma unroll 4
      for (; f + nsl < F; f += 2 * nsl) {         // two independent chains, loads unrolled
        s0 += x[f] * Wm[(size_t)f * C + j];
        s1 += x[f + nsl] * Wm[(size_t)(f + nsl) * C + j];
      }
      if (f < F) s0 += x[f] * Wm[(size_t)f * C + j];
    }
    __syncthreads();
    if (sl < nsl) partial[sl * C + j] = s0 + s1;
    __syncthreads();
    if (tid < C) {
      float r = b[tid];
      for (int q = 0; q < nsl; ++q) r += partial[q * C + tid];
      out[tid] = r;
    }
  } else {
    for (int j = tid; j < C; j += MT) {
      float s = b[j];
#pragma unroll 8
      for (int f = 0; f < F; ++f) s += x[f] * Wm[(size_t)f * C + j];
      out[j] = s;
    }
  }
}

// Same contraction with the [F][C] weight matrix first staged into LDS by the whole workgroup
// (independent 16-B loads: one or two memory round trips instead of F/nsl dependent ones).
// wl needs F*C floats.  Falls back to fc_small when the matrix does not fit `wl_cap` floats.
__device__ void fc_lds(const float* x, int F, const float* __restrict__ Wm,
                       const float* __restrict__ b, int C, float* __restrict__ out,
                       float* partial, float* wl, int wl_cap) {
  const int tid = threadIdx.x;
  const int tot = F * C;
  if (tot > wl_cap || C > MT) {
    fc_small(x, F, Wm, b, C, out, partial);
    return;
  }
  const int n4 = tot >> 2;
  const float4* W4 = reinterpret_cast<const float4*>(Wm);
  float4* wl4 = reinterpret_cast<float4*>(wl);
#pragma unroll 4
  for (int i = tid; i < n4; i += MT) wl4[i] = W4[i];
  for (int i = 4 * n4 + tid; i < tot; i += MT) wl[i] = Wm[i];
  __syncthreads();
  const int nsl = MT / C;
  const int j = tid % C, sl = tid / C;
  float s0 = 0.f, s1 = 0.f;
  if (sl < nsl) {
    int f = sl;
    for (; f + nsl < F; f += 2 * nsl) {
      s0 += x[f] * wl[f * C + j];
      s1 += x[f + nsl] * wl[(f + nsl) * C + j];
    }
    if (f < F) s0 += x[f] * wl[f * C + j];
    partial[sl * C + j] = s0 + s1;
  }
  __syncthreads();
  if (tid < C) {
    float r = b[tid];
    for (int q = 0; q < nsl; ++q) r += partial[q * C + tid];
    out[tid] = r;
  }
}

// ---------------------------------------------------------------------------------------------
// text maps: tmap[tslot, :] = word_vecs[t*N_full + n, :] . W_txt + b_txt
// A workgroup handles up to TM_GROUP nodes that share a weight set, so the [E, M] weight stream
// is read once per group.
// ---------------------------------------------------------------------------------------------
__device__ void textmap_item(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int item,
                             float* smem) {
  const int* tab = b.tab + tab_off + item * (2 + TM_GROUP);
  const int ws = tab[0], cnt = tab[1];
  const int E = b.E, Mp = b.Mp;
  float* wv = smem;                       // [TM_GROUP][E]
  float* part = wv + TM_GROUP * E;        // [4 waves][TM_GROUP][256]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // a wave copies whole word vectors: the node lookup (two dependent loads) once per row instead
  // of once per element in front of every load
  __shared__ int tslots[TM_GROUP];
  for (int g = wid; g < TM_GROUP; g += MT / 64) {
    const float* src = nullptr;
    if (g < cnt) {
      const DevNode& nd = b.nodes[tab[2 + g]];
      src = b.word_vecs + ((size_t)nd.t * b.N_full + nd.n) * E;    // nmn3_modules.py:53-57
      if (lane == 0) tslots[g] = nd.tslot;
    }
    for (int e = lane; e < E; e += 64) wv[g * E + e] = src ? src[e] : 0.f;
  }
  __syncthreads();
  // K-split over the 4 waves, float4 columns over the lanes: every lane streams its slice of
  // the padded [E][Mp] weight matrix with independent 16-B loads.
  const float4* Wp4 = reinterpret_cast<const float4*>(w.Wtxt[ws]);
  const float* bm = w.btxt[ws];
  const int eper = (E + 3) / 4;
  const int e0 = wid * eper, e1 = min(E, e0 + eper);
  for (int cb = 0; cb < Mp; cb += 256) {
    float4 acc[TM_GROUP];
#pragma unroll
    for (int g = 0; g < TM_GROUP; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* wp = Wp4 + (cb >> 2) + lane;
    constexpr int UE = 15;                  // weight rows of a lane in flight
    for (int eb = e0; eb < e1; eb += UE) {
      float4 w4[UE];
#pragma unroll
      for (int u = 0; u < UE; ++u) w4[u] = wp[(size_t)min(eb + u, e1 - 1) * (Mp >> 2)];
#pragma unroll
      for (int u = 0; u < UE; ++u) {
        const int e = eb + u;
        if (e < e1) {
#pragma unroll
          for (int g = 0; g < TM_GROUP; ++g) {
            const float x = wv[g * E + e];
            acc[g].x += x * w4[u].x; acc[g].y += x * w4[u].y; acc[g].z += x * w4[u].z;
            acc[g].w += x * w4[u].w;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < TM_GROUP; ++g)
      *reinterpret_cast<float4*>(part + ((size_t)(wid * TM_GROUP + g) * 256) + 4 * lane) = acc[g];
    __syncthreads();
    for (int i = tid; i < TM_GROUP * 256; i += MT) {
      const int g = i >> 8, c = i & 255;
      if (g < cnt && cb + c < Mp) {
        float r = bm[cb + c];
#pragma unroll
        for (int q = 0; q < 4; ++q) r += part[(size_t)(q * TM_GROUP + g) * 256 + c];
        b.tmap[(size_t)tslots[g] * Mp + cb + c] = r;
      }
    }
  }
}

// Every module kernel runs its work items in a loop: a host-scheduled launch (schedule.cpp) has one
// workgroup per item; a device-scheduled one (sched_kernel below: the tables and their lengths exist
// only in HBM) is a persistent grid that reads (offset, count) of launch slot `dl` from b.dsched.
#define N2_ITEM_LOOP(BODY)                                                        \
  extern __shared__ __attribute__((aligned(16))) float smem[];                   \
  if (dl >= 0) { tab_off = b.dsched[2 * dl]; count = b.dsched[2 * dl + 1]; }     \
  for (int item = blockIdx.x; item < count; item += gridDim.x) {                 \
    BODY(w, b, tab_off, item, smem);                                             \
    __syncthreads();                                                             \
  }

__global__ __launch_bounds__(MT) void textmap_kernel(ModuleWeights w, ModuleBuffers b, int tab_off,
                                                     int count, int dl) {
  N2_ITEM_LOOP(textmap_item)
}

// ---------------------------------------------------------------------------------------------
// Find-type epilogue on a hoisted conv_image map:
//   att[r] = l2norm_c( M[r, c] * tmap[c] (* amap[c]) ) . w_e + b_e   [ min with input_0 for Filter ]
// (nmn3_modules.py:104-108 Find, :129-130 Filter, :178-180 FindSameProperty)
// one wave per row, float4 lanes over the Mp channels; rows [r0, r1) of the map.
// ---------------------------------------------------------------------------------------------
__device__ void find_epilogue(const ModuleWeights& w, const ModuleBuffers& b, const DevNode& nd,
                              int node_id, int part, int nparts) {
  constexpr int MAXCI = 4;                 // Mp <= 1024
  const int HW = b.H * b.W, Mp = b.Mp;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool fsp = nd.op == N2NMN_OP_FIND_SAME_PROPERTY;
  const int wsel = fsp ? 1 : 0;
  const float* Mbuf = (fsp ? b.mfsp : b.mfind) + (size_t)nd.mslot * HW * Mp;
  const float* tm = b.tmap + (size_t)nd.tslot * Mp;
  const float be = w.be[wsel][0];
  const int rpp = (HW + nparts - 1) / nparts;
  const int r0 = part * rpp, r1 = min(HW, r0 + rpp);
  const float* in0 = (nd.op == N2NMN_OP_FILTER) ? b.arena + (size_t)nd.in0 * b.HWp : nullptr;
  float* outp = b.arena + (size_t)node_id * b.HWp;

  // row-invariant per-lane vectors: text map (times fc_att(att_feat) for FindSameProperty) and w_e
  float4 t4[MAXCI], e4[MAXCI];
#pragma unroll
  for (int i = 0; i < MAXCI; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < Mp) {
      t4[i] = *reinterpret_cast<const float4*>(tm + c);
      e4[i] = *reinterpret_cast<const float4*>(w.we[wsel] + c);   // padded to Mp
      if (fsp) {   // amap = fc_att(att_feat): bias + the POOL_PARTS partial sums of stage B
        float4 a4 = *reinterpret_cast<const float4*>(w.batt[0] + c);   // padded to Mp
        const float* pf = b.pfc + (size_t)nd.pslot * 2 * POOL_PARTS * Mp + c;
#pragma unroll
        for (int p = 0; p < POOL_PARTS; ++p) {
          const float4 q = *reinterpret_cast<const float4*>(pf + p * Mp);
          a4.x += q.x; a4.y += q.y; a4.z += q.z; a4.w += q.w;
        }
        t4[i].x *= a4.x; t4[i].y *= a4.y; t4[i].z *= a4.z; t4[i].w *= a4.w;
      }
    }
  }
  constexpr int UNR = 5;
  for (int rb = r0 + wid; rb < r1; rb += UNR * (MT / 64)) {
    float ss[UNR], dot[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + u * (MT / 64);
      ss[u] = 0.f; dot[u] = 0.f;
      if (r < r1) {
#pragma unroll
        for (int i = 0; i < MAXCI; ++i) {
          const int c = 4 * lane + 256 * i;
          if (c < Mp) {
            const float4 m4 = *reinterpret_cast<const float4*>(Mbuf + (size_t)r * Mp + c);
            const float p0 = m4.x * t4[i].x, p1 = m4.y * t4[i].y, p2 = m4.z * t4[i].z,
                        p3 = m4.w * t4[i].w;
            ss[u] += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
            dot[u] += p0 * e4[i].x + p1 * e4[i].y + p2 * e4[i].z + p3 * e4[i].w;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + u * (MT / 64);
      const float s2 = wave_sum(ss[u]);
      const float d2 = wave_sum(dot[u]);
      if (lane == 0 && r < r1) {
        float att = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;    // tf.nn.l2_normalize eps (A.4)
        if (in0) att = fminf(in0[r], att);                 // Filter = And(input_0, Find)
        outp[r] = att;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Transform (nmn3_modules.py:185-216): conv KSxKS SAME of the 1-channel attention map to M
// channels, times the text map, l2-normalise over channels, dot with w_e.
// Lanes = output pixels (each lane keeps its KSxKS input window in registers), waves = channel
// quarters; the filter taps, pre-multiplied by the text map (K'[c][tap] = K[tap][c]*t[c], with
// b'[c] and w_e[c] appended), sit in LDS and are read with wave-uniform (broadcast) 16-B reads.
// The channel reduction is lane-local, so there is no cross-lane traffic in the hot loop.
// ---------------------------------------------------------------------------------------------
template <int KS>
__device__ void transform_op(const ModuleWeights& w, const ModuleBuffers& b, const DevNode& nd,
                             int node_id, int part, int nparts, float* smem) {
  constexpr int KK = KS * KS;
  constexpr int RS = (KK + 2 + 3) & ~3;      // row stride of the tap table (floats, 16-B multiple)
  constexpr int PAD = KS / 2;
  const int H = b.H, W = b.W, HW = H * W, M = b.M, Mp = b.Mp;
  const int PW = W + 2 * PAD, PH = H + 2 * PAD;
  float* Kl = smem;                            // [M][RS]
  float* xin = Kl + (size_t)M * RS;            // [PH][PW]
  float* red = xin + ((PH * PW + 3) & ~3);     // [4 waves][64][2]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* in0 = b.arena + (size_t)nd.in0 * b.HWp;
  const float* tm = b.tmap + (size_t)nd.tslot * Mp;
  for (int i = tid; i < PH * PW; i += MT) {
    const int y = i / PW - PAD, x = i % PW - PAD;
    xin[i] = (y >= 0 && y < H && x >= 0 && x < W) ? in0[y * W + x] : 0.f;
  }
#pragma unroll 4
  for (int i = tid; i < KK * M; i += MT) {     // coalesced over c, independent loads
    const int tap = i / M, c = i - tap * M;
    Kl[c * RS + tap] = w.Kt[i] * tm[c];
  }
  for (int c = tid; c < M; c += MT) {
    Kl[c * RS + KK] = w.bt[c] * tm[c];
    Kl[c * RS + KK + 1] = w.we[2][c];
  }
  __syncthreads();
  const float be = w.be[2][0];
  float* outp = b.arena + (size_t)node_id * b.HWp;
  const int ppp = (HW + nparts - 1) / nparts;
  const int p0 = part * ppp, p1 = min(HW, p0 + ppp);
  for (int pb = p0; pb < p1; pb += 64) {
    const int p = pb + lane;
    const bool on = p < p1;
    const int y = on ? p / W : 0, x = on ? p - (p / W) * W : 0;
    float win[KK];
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) win[dy * KS + dx] = xin[(y + dy) * PW + x + dx];
    float ss = 0.f, dot = 0.f;
    for (int c = wid; c < M; c += MT / 64) {
      const float4* kr = reinterpret_cast<const float4*>(Kl + (size_t)c * RS);
      float k[RS];
#pragma unroll
      for (int q = 0; q < RS / 4; ++q) {
        const float4 t = kr[q];
        k[4 * q] = t.x; k[4 * q + 1] = t.y; k[4 * q + 2] = t.z; k[4 * q + 3] = t.w;
      }
      float v = k[KK];
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) v += k[tap] * win[tap];
      ss += v * v;
      dot += v * k[KK + 1];
    }
    __syncthreads();
    red[(wid * 64 + lane) * 2] = ss;
    red[(wid * 64 + lane) * 2 + 1] = dot;
    __syncthreads();
    if (wid == 0 && on) {
      float s2 = 0.f, d2 = 0.f;
#pragma unroll
      for (int q = 0; q < MT / 64; ++q) { s2 += red[(q * 64 + lane) * 2]; d2 += red[(q * 64 + lane) * 2 + 1]; }
      outp[p] = d2 / sqrtf(fmaxf(s2, 1e-12f)) + be;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// answer heads on raw attention maps: Exist (:258-280), Count (:282-304),
// EqualNum / MoreNum / LessNum (:306-400)
// ---------------------------------------------------------------------------------------------
// LDS floats the raw-map answer heads may use to stage their fc weights (host and device agree)
__host__ __device__ inline int light_wl_cap(int HW, int C) {
  const int full = (2 * HW + 4) * C;
  return full < 12288 ? full : 12288;
}

__device__ void light_answer(const ModuleWeights& w, const ModuleBuffers& b, const DevNode& nd,
                             float* smem) {
  const int HW = b.H * b.W, C = b.C;
  float* x = smem;                    // up to 2*HW + 4 features
  float* scratch = x + ((2 * HW + 4 + 3) & ~3);    // 16 floats for block reductions
  float* partial = scratch + 16;      // 256 floats
  float* wl = partial + 256;          // staged fc weights
  const int nin = (nd.op == N2NMN_OP_EXIST || nd.op == N2NMN_OP_COUNT) ? 1 : 2;
  const int tid = threadIdx.x;
  float mn[2], mx[2], sm[2];
  for (int i = 0; i < nin; ++i) {
    const float* src = b.arena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    float lmn = INFINITY, lmx = -INFINITY, lsm = 0.f;
    for (int r = tid; r < HW; r += MT) {
      const float v = src[r];
      x[i * (HW + 2) + r] = v;        // row-major y*W + x flatten (:297)
      lmn = fminf(lmn, v); lmx = fmaxf(lmx, v); lsm += v;
    }
    mn[i] = block_reduce<2>(lmn, scratch);
    mx[i] = block_reduce<1>(lmx, scratch);
    sm[i] = block_reduce<0>(lsm, scratch);
  }
  __syncthreads();
  int F, wi;
  if (nd.op == N2NMN_OP_EXIST) {
    if (tid == 0) { x[0] = mn[0]; x[1] = sm[0] / (float)HW; x[2] = mx[0]; }
    F = 3; wi = 0;
  } else if (nd.op == N2NMN_OP_COUNT) {
    if (tid == 0) { x[HW] = mn[0]; x[HW + 1] = mx[0]; }
    F = HW + 2; wi = 1;
  } else {
    if (tid == 0) {
      x[HW] = mn[0]; x[HW + 1] = mx[0];
      x[2 * HW + 2] = mn[1]; x[2 * HW + 3] = mx[1];
    }
    F = 2 * HW + 4;
    wi = nd.op == N2NMN_OP_EQUAL_NUM ? 2 : (nd.op == N2NMN_OP_MORE_NUM ? 3 : 4);
  }
  __syncthreads();
  fc_lds(x, F, w.Wans[wi], w.bans[wi], C, b.scores + (size_t)nd.out_row * C, partial, wl,
         light_wl_cap(HW, C));
}

__device__ void att_ops_item(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int item,
                             float* smem) {
  const int* e = b.tab + tab_off + item * 4;
  const int node_id = e[0], part = e[1], nparts = e[2];
  const DevNode nd = b.nodes[node_id];
  const int HW = b.H * b.W;
  float* outp = b.arena + (size_t)node_id * b.HWp;
  switch (nd.op) {
    case N2NMN_OP_SCENE:                                   // :60-72  att = 3.0 everywhere
      for (int r = threadIdx.x; r < HW; r += MT) outp[r] = 3.0f;
      break;
    case N2NMN_OP_FIND:
    case N2NMN_OP_FILTER:
    case N2NMN_OP_FIND_SAME_PROPERTY:
      find_epilogue(w, b, nd, node_id, part, nparts);
      break;
    case N2NMN_OP_TRANSFORM:
      if (b.ksize == 5) transform_op<5>(w, b, nd, node_id, part, nparts, smem);
      else transform_op<3>(w, b, nd, node_id, part, nparts, smem);
      break;
    case N2NMN_OP_AND:                                     // :218-236
    case N2NMN_OP_OR: {                                    // :238-256
      const float* a0 = b.arena + (size_t)nd.in0 * b.HWp;
      const float* a1 = b.arena + (size_t)nd.in1 * b.HWp;
      for (int r = threadIdx.x; r < HW; r += MT)
        outp[r] = nd.op == N2NMN_OP_AND ? fminf(a0[r], a1[r]) : fmaxf(a0[r], a1[r]);
      break;
    }
    default:
      light_answer(w, b, nd, smem);
      break;
  }
}

__global__ __launch_bounds__(MT) void att_ops_kernel(ModuleWeights w, ModuleBuffers b, int tab_off,
                                                     int count, int dl) {
  N2_ITEM_LOOP(att_ops_item)
}

// ---------------------------------------------------------------------------------------------
// pool: a = softmax_HW(att logits);  f = sum_hw a[hw] * feat[n, hw, :]  (the HBM-bound read of
// the [H*W, D] feature map), then this part's slice of fc_att:  pfc[part, c] = f[c0:c0+Dp] .
// W_att[c0:c0+Dp, c].  One workgroup per (node, channel part): POOL_PARTS x nodes workgroups fill
// the chip; SameProperty pools both inputs from ONE feature read.  Each thread owns a float4
// channel column and strides over rows (8 rows in flight per workgroup -> ~19 independent 16-B
// loads per thread).
// ---------------------------------------------------------------------------------------------
__device__ void pool_item(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int item,
                          float* smem) {
  const int* e = b.tab + tab_off + item * 2;
  const int node_id = e[0], part = e[1];
  const DevNode nd = b.nodes[node_id];
  const int HW = b.H * b.W, D = b.D, Mp = b.Mp;
  const int Dp = D / POOL_PARTS, c0 = part * Dp;
  const int tid = threadIdx.x;
  const int nin = nd.op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
  float* a0 = smem;                    // [HW]
  float* a1 = a0 + ((HW + 3) & ~3);    // [HW]
  float* scratch = a1 + ((HW + 3) & ~3);   // [16]
  float* pooled = scratch + 16;        // [2][Dp]
  float* stage = pooled + 2 * Dp;      // [rows in flight][2][Dp]

  const int ncol = Dp / 4;             // float4 columns of this part
  const int nrow = MT / ncol;          // rows in flight
  const int lc = tid % ncol, lr = tid / ncol;
  // The feature rows of this thread do not depend on the softmax: issue all of their 16-B loads
  // first so the HBM round trip overlaps the softmax prologue (register path for <= PR rows).
  constexpr int PR = 20;
  const int myrows = lr < nrow ? (HW - lr + nrow - 1) / nrow : 0;
  const bool regpath = (HW + nrow - 1) / nrow <= PR;
  const float* fp = b.feat + (size_t)nd.n * HW * D + c0 + 4 * lc;
  float4 fr[PR];
  if (regpath) {
#pragma unroll
    for (int q = 0; q < PR; ++q) {
      const int r = lr + q * nrow;
      fr[q] = *reinterpret_cast<const float4*>(fp + (size_t)(q < myrows ? r : lr) * D);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  for (int i = 0; i < nin; ++i) {      // softmax over the H*W logits (:170-172,432-437,482-484)
    const float* src = b.arena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    float* dst = i == 0 ? a0 : a1;
    float lm = -INFINITY;
    for (int r = tid; r < HW; r += MT) lm = fmaxf(lm, src[r]);
    const float mx = block_reduce<1>(lm, scratch);
    float ls = 0.f;
    for (int r = tid; r < HW; r += MT) {
      const float ex = expf(src[r] - mx);
      dst[r] = ex;
      ls += ex;
    }
    const float sum = block_reduce<0>(ls, scratch);
    for (int r = tid; r < HW; r += MT) dst[r] = dst[r] / sum;
  }
  __syncthreads();

  float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
  if (lr < nrow) {
    if (regpath) {
#pragma unroll
      for (int q = 0; q < PR; ++q) {
        if (q < myrows) {
          const int r = lr + q * nrow;
          const float w0 = a0[r], w1 = nin == 2 ? a1[r] : 0.f;
          acc0.x += w0 * fr[q].x; acc0.y += w0 * fr[q].y; acc0.z += w0 * fr[q].z; acc0.w += w0 * fr[q].w;
          acc1.x += w1 * fr[q].x; acc1.y += w1 * fr[q].y; acc1.z += w1 * fr[q].z; acc1.w += w1 * fr[q].w;
        }
      }
    } else {
#pragma unroll 8
      for (int r = lr; r < HW; r += nrow) {
        const float4 f4 = *reinterpret_cast<const float4*>(fp + (size_t)r * D);
        const float w0 = a0[r], w1 = nin == 2 ? a1[r] : 0.f;
        acc0.x += w0 * f4.x; acc0.y += w0 * f4.y; acc0.z += w0 * f4.z; acc0.w += w0 * f4.w;
        acc1.x += w1 * f4.x; acc1.y += w1 * f4.y; acc1.z += w1 * f4.z; acc1.w += w1 * f4.w;
      }
    }
    *reinterpret_cast<float4*>(stage + (size_t)(lr * 2 + 0) * Dp + 4 * lc) = acc0;
    *reinterpret_cast<float4*>(stage + (size_t)(lr * 2 + 1) * Dp + 4 * lc) = acc1;
  }
  __syncthreads();
  for (int i = tid; i < 2 * Dp; i += MT) {
    float s = 0.f;
    for (int q = 0; q < nrow; ++q) s += stage[(size_t)q * 2 * Dp + i];
    pooled[i] = s;
    if (b.pooled)                        // training: [pslot][input][D]
      b.pooled[((size_t)nd.pslot * 2 + i / Dp) * D + c0 + (i % Dp)] = s;
  }
  __syncthreads();

  // partial fc_att over this part's channels: K-split over the 4 waves, float4 columns over
  // the lanes of the padded [D][Mp] weight matrix
  float* fpart = stage;                // [4 waves][256] (stage is free again)
  const int lane = tid & 63, wid = tid >> 6;
  const int kper = (Dp + 3) / 4;
  const int k0 = wid * kper, k1 = min(Dp, k0 + kper);
  for (int i = 0; i < nin; ++i) {
    int wi;
    if (nd.op == N2NMN_OP_FIND_SAME_PROPERTY) wi = 0;
    else if (nd.op == N2NMN_OP_SAME_PROPERTY) wi = 1 + i;
    else wi = 3;
    const float4* Wp4 = reinterpret_cast<const float4*>(w.Watt[wi] + (size_t)c0 * Mp);
    const float* pv = pooled + i * Dp;
    float* dst = b.pfc + (((size_t)nd.pslot * 2 + i) * POOL_PARTS + part) * Mp;
    for (int cb = 0; cb < Mp; cb += 256) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* wp = Wp4 + (cb >> 2) + lane;
#pragma unroll 16
      for (int k = k0; k < k1; ++k) {
        const float4 w4 = wp[(size_t)k * (Mp >> 2)];
        const float x = pv[k];
        acc.x += x * w4.x; acc.y += x * w4.y; acc.z += x * w4.z; acc.w += x * w4.w;
      }
      __syncthreads();
      *reinterpret_cast<float4*>(fpart + wid * 256 + 4 * lane) = acc;
      __syncthreads();
      if (cb + tid < Mp)
        dst[cb + tid] = fpart[tid] + fpart[256 + tid] + fpart[512 + tid] + fpart[768 + tid];
    }
  }
}

__global__ __launch_bounds__(MT) void pool_kernel(ModuleWeights w, ModuleBuffers b, int tab_off, int count,
                                                  int dl) {
  N2_ITEM_LOOP(pool_item)
}

// ---------------------------------------------------------------------------------------------
// heads: Describe  scores = l2norm(tmap * a) . W_e + b          (:479-493)
//        SameProperty  scores = l2norm(a0 * tmap * a1) . W_e + b (:424-450)
// a = b_att + sum of the POOL_PARTS partial fc_att rows of stage B.
// ---------------------------------------------------------------------------------------------
__device__ void heads_item(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int item,
                           float* smem) {
  const int node_id = b.tab[tab_off + item];
  const DevNode nd = b.nodes[node_id];
  const int M = b.M, Mp = b.Mp, C = b.C;
  float* ev = smem;                 // [Mp]
  float* scratch = ev + Mp;         // [16]
  float* partial = scratch + 16;    // [256]
  float* wl = partial + 256;        // staged fc_eltwise weights
  const bool same = nd.op == N2NMN_OP_SAME_PROPERTY;
  const float* tm = b.tmap + (size_t)nd.tslot * Mp;
  const float* pf = b.pfc + (size_t)nd.pslot * 2 * POOL_PARTS * Mp;
  float lss = 0.f;
  for (int c = threadIdx.x; c < Mp; c += MT) {
    float v = 0.f;
    if (c < M) {
      float a0 = w.batt[same ? 1 : 3][c];
      for (int p = 0; p < POOL_PARTS; ++p) a0 += pf[p * Mp + c];
      v = a0 * tm[c];
      if (same) {
        float a1 = w.batt[2][c];
        for (int p = 0; p < POOL_PARTS; ++p) a1 += pf[(POOL_PARTS + p) * Mp + c];
        v *= a1;
      }
    }
    ev[c] = v;
    lss += v * v;
  }
  const float ss = block_reduce<0>(lss, scratch);
  const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  if (b.ev_out) {          // large answer vocabulary: fc_eltwise runs as a GEMM over the launch
    // (device-scheduled: ONE GEMM over the questions after the last level, row = question; the row
    // lists were written by sched_kernel)
    const int er = b.ev_by_q ? nd.out_row : item;
    for (int c = threadIdx.x; c < Mp; c += MT) b.ev_out[(size_t)er * Mp + c] = ev[c] * inv;
    if (threadIdx.x == 0 && !b.ev_by_q) {
      b.ev_rows[item] = same ? -1 : nd.out_row;
      b.ev_rows[b.ev_stride + item] = same ? nd.out_row : -1;
    }
    return;
  }
  for (int c = threadIdx.x; c < Mp; c += MT) ev[c] *= inv;
  __syncthreads();
  const int wi = same ? 5 : 6;
  fc_lds(ev, M, w.Wans[wi], w.bans[wi], C, b.scores + (size_t)nd.out_row * C, partial, wl,
         b.wl_cap);
}

__global__ __launch_bounds__(MT) void heads_kernel(ModuleWeights w, ModuleBuffers b, int tab_off, int count,
                                                   int dl) {
  N2_ITEM_LOOP(heads_item)
}

// models_vqa/nmn3_modules.py:11-31: tf.linspace(-1., 1., n)[i] = -1 + i * (2 / (n - 1))
__global__ void add_coords_kernel(const float* __restrict__ feat, int N, int H, int W, int D0,
                                  int D, float* __restrict__ out) {
  const size_t total = (size_t)N * H * W * D;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    const size_t px = i / D;
    float v = 0.f;
    if (c < D0) v = feat[px * D0 + c];
    else if (c == D0) { const int x = (int)(px % W); v = W > 1 ? -1.f + x * (2.f / (W - 1)) : -1.f; }
    else if (c == D0 + 1) { const int y = (int)((px / W) % H); v = H > 1 ? -1.f + y * (2.f / (H - 1)) : -1.f; }
    out[i] = v;
  }
}


// ---------------------------------------------------------------------------------------------
// sched_kernel: the layout assembler and the level scheduler on the device (SchedArgs).  ONE
// workgroup; thread i decodes questions i, i + 1024, ...  Three passes with the counters in LDS:
// count the work items of every (stage, level) and text-map weight set, prefix them into table
// offsets (thread 0), place the items with LDS cursors.  The order of the items inside a table is
// whatever the atomics give; no result depends on it (every item writes its own rows).
// ---------------------------------------------------------------------------------------------
constexpr int SCHED_THREADS = 1024;

__device__ __forceinline__ int dev_arity(int op) {        // nmn3_assembler.py:9-24 (schedule.cpp op_arity)
  switch (op) {
    case N2NMN_OP_SCENE: case N2NMN_OP_FIND: return 0;
    case N2NMN_OP_FILTER: case N2NMN_OP_FIND_SAME_PROPERTY: case N2NMN_OP_TRANSFORM:
    case N2NMN_OP_EXIST: case N2NMN_OP_COUNT: case N2NMN_OP_DESCRIBE: return 1;
    case N2NMN_OP_AND: case N2NMN_OP_OR: case N2NMN_OP_EQUAL_NUM: case N2NMN_OP_MORE_NUM:
    case N2NMN_OP_LESS_NUM: case N2NMN_OP_SAME_PROPERTY: return 2;
    default: return -1;
  }
}
__device__ __forceinline__ bool dev_is_answer(int op) {   // nmn3_assembler.py:26-41
  return op == N2NMN_OP_EXIST || op == N2NMN_OP_COUNT || op == N2NMN_OP_EQUAL_NUM ||
         op == N2NMN_OP_MORE_NUM || op == N2NMN_OP_LESS_NUM || op == N2NMN_OP_SAME_PROPERTY ||
         op == N2NMN_OP_DESCRIBE;
}
__device__ __forceinline__ bool dev_is_pool(int op) {
  return op == N2NMN_OP_FIND_SAME_PROPERTY || op == N2NMN_OP_SAME_PROPERTY || op == N2NMN_OP_DESCRIBE;
}
__device__ __forceinline__ int dev_text_set(int op) {     // schedule.cpp text_weight_set
  switch (op) {
    case N2NMN_OP_FIND: case N2NMN_OP_FILTER: return 0;
    case N2NMN_OP_FIND_SAME_PROPERTY: return 1;
    case N2NMN_OP_TRANSFORM: return 2;
    case N2NMN_OP_SAME_PROPERTY: return 3;
    case N2NMN_OP_DESCRIBE: return 4;
    default: return -1;
  }
}

struct SchedLayout {          // one question's decoded layout (thread-private)
  int nn;                     // nodes are the token positions [0, nn) when valid
  bool valid;
  int8_t op[SCHED_MAX_T], in0[SCHED_MAX_T], in1[SCHED_MAX_T], level[SCHED_MAX_T];
};

// nmn3_assembler.py:153-222 / schedule.cpp assemble_tokens + the level rule of schedule()
__device__ void sched_decode(const SchedArgs& a, int n, SchedLayout& q) {
  q.nn = 0; q.valid = false;
  bool has_eos = false;
  for (int t = 0; t < a.T; ++t) {
    const int tok = a.tokens[(size_t)t * a.N + n];
    if (tok < 0 || tok >= a.V) return;                     // garbage token: not a layout (as the walker)
    if (a.token_op[tok] < 0) has_eos = true;
  }
  if (!has_eos) return;                                    // :172-173
  int8_t stack[SCHED_MAX_T], outl[SCHED_MAX_T];
  int sp = 0, nn = 0;
  for (int t = 0; t < a.T; ++t) {
    const int op = a.token_op[a.tokens[(size_t)t * a.N + n]];
    if (op < 0) break;                                     // <eos>
    const int k = dev_arity(op);
    if (k < 0 || sp < k) return;                           // :189-191
    int in0 = -1, in1 = -1, in_max = -1;
    for (int j = k - 1; j >= 0; --j) {                     // :194-199 input_{k-1} = stack top
      const int top = stack[--sp];
      if (dev_is_answer(q.op[top])) return;
      (j == 0 ? in0 : in1) = top;
      in_max = max(in_max, (int)outl[top]);
    }
    int lvl;
    if (k == 0) { lvl = 0; outl[t] = 0; }
    else if (dev_is_pool(op)) { lvl = max(in_max, 0); outl[t] = op == N2NMN_OP_FIND_SAME_PROPERTY ? lvl + 1 : lvl; }
    else { lvl = in_max + 1; outl[t] = lvl; }
    q.op[t] = (int8_t)op; q.in0[t] = (int8_t)in0; q.in1[t] = (int8_t)in1; q.level[t] = (int8_t)lvl;
    stack[sp++] = (int8_t)t;
    nn = t + 1;
  }
  if (sp != 1 || !dev_is_answer(q.op[stack[0]])) return;   // :205-211
  q.nn = nn; q.valid = true;
}

__global__ __launch_bounds__(SCHED_THREADS) void sched_kernel(SchedArgs a) {
  __shared__ int cnt[3][SCHED_MAX_T + 2], off[3][SCHED_MAX_T + 2], cur[3][SCHED_MAX_T + 2];
  __shared__ int tcnt[5], tbase[5], tcur[5], ngroups, fits;
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * (SCHED_MAX_T + 2); i += SCHED_THREADS) { (&cnt[0][0])[i] = 0; (&cur[0][0])[i] = 0; }
  if (tid < 5) { tcnt[tid] = 0; tcur[tid] = 0; }
  __syncthreads();
  // ---- pass 1: nodes, validity, counts --------------------------------------------------------------
  SchedLayout q;
  for (int n = tid; n < a.N; n += SCHED_THREADS) {
    sched_decode(a, n, q);
    if (a.validity) a.validity[n] = q.valid ? 1 : 0;
    int root_op = -1;
    for (int t = 0; t < a.T; ++t) {
      DevNode d;
      d.op = -1; d.t = t; d.n = n; d.in0 = d.in1 = -1; d.out_row = -1;
      d.tslot = d.pslot = d.mslot = -1; d.level = -1;
#pragma unroll
      for (int i = 0; i < 6; ++i) d.pad[i] = 0;
      if (q.valid && t < q.nn) {
        const int op = q.op[t], lvl = q.level[t], id0 = n * a.T;
        d.op = op; d.level = lvl;
        d.in0 = q.in0[t] >= 0 ? id0 + q.in0[t] : -1;
        d.in1 = q.in1[t] >= 0 ? id0 + q.in1[t] : -1;
        d.out_row = t == q.nn - 1 ? n : -1;               // the last node is the root (stack size 1)
        if (t == q.nn - 1) root_op = op;
        const int ws = dev_text_set(op);
        if (ws >= 0) { d.tslot = id0 + t; atomicAdd(&tcnt[ws], 1); }
        if (dev_is_pool(op)) d.pslot = id0 + t;
        if (op == N2NMN_OP_FIND || op == N2NMN_OP_FILTER || op == N2NMN_OP_FIND_SAME_PROPERTY) d.mslot = n;
        // stage A items
        if (op == N2NMN_OP_FIND_SAME_PROPERTY) atomicAdd(&cnt[0][lvl + 1], FIND_PARTS);
        else if (!dev_is_pool(op))
          atomicAdd(&cnt[0][lvl], (op == N2NMN_OP_FIND || op == N2NMN_OP_FILTER) ? FIND_PARTS
                                  : op == N2NMN_OP_TRANSFORM ? TRANSFORM_PARTS : 1);
        if (dev_is_pool(op)) atomicAdd(&cnt[1][lvl], POOL_PARTS);
        if (op == N2NMN_OP_DESCRIBE || op == N2NMN_OP_SAME_PROPERTY) atomicAdd(&cnt[2][lvl], 1);
      }
      a.nodes[(size_t)n * a.T + t] = d;
    }
    if (a.ev_rows) {
      a.ev_rows[n] = root_op == N2NMN_OP_DESCRIBE ? n : -1;
      a.ev_rows[a.ev_stride + n] = root_op == N2NMN_OP_SAME_PROPERTY ? n : -1;
    }
  }
  __syncthreads();
  // ---- offsets ------------------------------------------------------------------------------------------
  if (tid == 0) {
    int g = 0;
    for (int ws = 0; ws < 5; ++ws) { tbase[ws] = g; g += (tcnt[ws] + TM_GROUP - 1) / TM_GROUP; }
    ngroups = g;
    int o = g * (2 + TM_GROUP);
    bool ok = true;
    for (int l = 0; l <= SCHED_MAX_T; ++l) {
      off[0][l] = o; o += 4 * cnt[0][l];
      off[1][l] = o; o += 2 * cnt[1][l];
      off[2][l] = o; o += cnt[2][l];
      if (l >= a.levels && (cnt[0][l] | cnt[1][l] | cnt[2][l])) ok = false;   // deeper than the launches
    }
    if (o > a.tab_cap) ok = false;
    fits = ok;
    if (!ok && a.overflow) *a.overflow = 1;
    a.dsched[0] = 0; a.dsched[1] = ok ? g : 0;
    for (int l = 0; l < a.levels; ++l)
      for (int st = 0; st < 3; ++st) {
        a.dsched[2 * (1 + 3 * l + st)] = off[st][l];
        a.dsched[2 * (1 + 3 * l + st) + 1] = ok ? cnt[st][l] : 0;
      }
  }
  __syncthreads();
  if (!fits) {
    // the tables do not fit (unreachable with the context's worst-case sizing, capi.cpp max_tab): nothing is
    // launched, so no question of the launch has an answer -- say so instead of returning zero logits as valid
    if (a.validity)
      for (int n = tid; n < a.N; n += SCHED_THREADS) a.validity[n] = 0;
    return;
  }
  // ---- pass 2: place ----------------------------------------------------------------------------------------
  for (int n = tid; n < a.N; n += SCHED_THREADS) {
    if (a.N > SCHED_THREADS) sched_decode(a, n, q);       // (one question per thread: still in registers)
    if (!q.valid) continue;
    for (int t = 0; t < q.nn; ++t) {
      const int op = q.op[t], lvl = q.level[t], id = n * a.T + t;
      const int ws = dev_text_set(op);
      if (ws >= 0) {
        const int k = atomicAdd(&tcur[ws], 1);
        a.tab[(tbase[ws] + k / TM_GROUP) * (2 + TM_GROUP) + 2 + k % TM_GROUP] = id;
      }
      int parts = 0, la = lvl;
      if (op == N2NMN_OP_FIND_SAME_PROPERTY) { parts = FIND_PARTS; la = lvl + 1; }
      else if (!dev_is_pool(op))
        parts = (op == N2NMN_OP_FIND || op == N2NMN_OP_FILTER) ? FIND_PARTS
                : op == N2NMN_OP_TRANSFORM ? TRANSFORM_PARTS : 1;
      if (parts) {
        const int k = atomicAdd(&cur[0][la], parts);
        for (int p = 0; p < parts; ++p) {
          int* e = a.tab + off[0][la] + 4 * (k + p);
          e[0] = id; e[1] = p; e[2] = parts; e[3] = 0;
        }
      }
      if (dev_is_pool(op)) {
        const int k = atomicAdd(&cur[1][lvl], POOL_PARTS);
        for (int p = 0; p < POOL_PARTS; ++p) {
          int* e = a.tab + off[1][lvl] + 2 * (k + p);
          e[0] = id; e[1] = p;
        }
      }
      if (op == N2NMN_OP_DESCRIBE || op == N2NMN_OP_SAME_PROPERTY)
        a.tab[off[2][lvl] + atomicAdd(&cur[2][lvl], 1)] = id;
    }
  }
  __syncthreads();
  // text-map group headers: [weight set, nodes in the group], unused node slots = -1
  for (int g = tid; g < ngroups; g += SCHED_THREADS) {
    int ws = 0;
    while (ws < 4 && g >= tbase[ws + 1]) ++ws;
    const int c = min(TM_GROUP, tcnt[ws] - TM_GROUP * (g - tbase[ws]));
    int* e = a.tab + g * (2 + TM_GROUP);
    e[0] = ws; e[1] = c;
    for (int j = c; j < TM_GROUP; ++j) e[2 + j] = -1;
  }
}

}  // namespace

void launch_add_coords(const float* feat, int N, int H, int W, int D0, int D, float* out,
                       hipStream_t s) {
  const size_t total = (size_t)N * H * W * D;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(add_coords_kernel, dim3(blocks), dim3(256), 0, s, feat, N, H, W, D0, D, out);
}

void launch_sched(const SchedArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(sched_kernel, dim3(1), dim3(SCHED_THREADS), 0, s, a);
}

void launch_textmap(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                    hipStream_t s, int dl) {
  const size_t smem = sizeof(float) * ((size_t)TM_GROUP * b.E + 4 * TM_GROUP * 256);
  hipLaunchKernelGGL(textmap_kernel, dim3(count), dim3(MT), smem, s, w, b, tab_off, count, dl);
}

void launch_att_ops(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                    hipStream_t s, int dl) {
  const int HW = b.H * b.W;
  const int pad = b.ksize / 2;
  const int KK = b.ksize * b.ksize;
  const int RS = (KK + 2 + 3) & ~3;
  const size_t tr = (size_t)b.M * RS + (((size_t)(b.H + 2 * pad) * (b.W + 2 * pad) + 3) & ~3) +
                    4 * 64 * 2;
  const size_t la = (size_t)((2 * HW + 4 + 3) & ~3) + 16 + 256 + (size_t)light_wl_cap(HW, b.C);
  // models_vqa has neither the conv Transform nor the raw-map answer heads: Find-type epilogues,
  // And/Or and Scene use no dynamic LDS, so many workgroups fit a CU
  const size_t smem = b.vqa ? 0 : sizeof(float) * (tr > la ? tr : la);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(att_ops_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(att_ops_kernel, dim3(count), dim3(MT), smem, s, w, b, tab_off, count, dl);
}

void launch_pool(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                 hipStream_t s, int dl) {
  const int HW = b.H * b.W, Dp = b.D / POOL_PARTS;
  const int nrow = MT / (Dp / 4);
  const size_t stage = std::max<size_t>((size_t)nrow * 2 * Dp, 1024);
  const size_t smem = sizeof(float) * (2 * (size_t)((HW + 3) & ~3) + 16 + 2 * Dp + stage);
  hipLaunchKernelGGL(pool_kernel, dim3(count), dim3(MT), smem, s, w, b, tab_off, count, dl);
}

void launch_heads(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                  hipStream_t s, int dl) {
  const size_t smem = sizeof(float) * ((size_t)b.Mp + 16 + 256 + (size_t)b.wl_cap);
  hipLaunchKernelGGL(heads_kernel, dim3(count), dim3(MT), smem, s, w, b, tab_off, count, dl);
}

}  // namespace n2nmn
