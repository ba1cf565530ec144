// Backward of the module operators (models_clevr/nmn3_modules.py:60-495), run level by level in
// the reverse order of the forward stages (schedule.cpp): heads -> pool -> att of each level, then
// the hoisted conv_image weight gradients (gemm_tn) and the text maps.
//
// Every attention node of a layout tree has exactly one consumer, so garena[node] (d loss / d
// attention map) is written once, by that consumer.  Parameter gradients accumulate atomically
// into the zeroed flat gradient buffer; gradients of the hoisted conv_image maps accumulate per
// image slot (several Find / Filter nodes may share one image) and become dW through gemm_tn.
//
// TF gradient conventions (TF 1.0.0 math_grad.py) restated:
//   tf.minimum / maximum : ties -> first argument          (And / Or / Filter)
//   tf.reduce_min / max  : split equally between ties      (Exist / Count / *Num)
//   tf.nn.l2_normalize   : y = x * rsqrt(max(ss, eps)) differentiated as written
#include <algorithm>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

constexpr int MT = 256;

// d of y = x * rsqrt(max(ss, 1e-12)):  dx = inv*dy - (ss > eps ? x * inv^3 * (x . dy) : 0)
// callers apply it with their own layouts; this is only the scalar factor of the second term.
__device__ __forceinline__ float l2n_k(float ss, float inv) {
  return ss > 1e-12f ? inv * inv * inv : 0.f;
}

// ---------------------------------------------------------------------------------------------
// heads_bwd: Describe  scores = l2n(tm * a0) . W_e + b           (:479-493)
//            SameProperty  scores = l2n(a0 * tm * a1) . W_e + b  (:424-450)
// one workgroup per node.  Outputs dtmap[tslot], dpfc[pslot][0/1] (= d fc_att outputs), and the
// fc_eltwise / fc_att-bias gradients.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MT) void heads_bwd_kernel(ModuleWeights w, ModuleBuffers b,
                                                       ModuleGrads g, int tab_off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int node_id = b.tab[tab_off + blockIdx.x];
  const DevNode nd = b.nodes[node_id];
  const int M = b.M, Mp = b.Mp, C = b.C;
  float* a0s = smem;              // [Mp]
  float* a1s = a0s + Mp;          // [Mp]
  float* tms = a1s + Mp;          // [Mp]
  float* evs = tms + Mp;          // [Mp] un-normalised product, later d ev
  float* ds = evs + Mp;           // [C padded to 32]
  float* scratch = ds + ((C + 31) & ~31);   // [16]
  const bool same = nd.op == N2NMN_OP_SAME_PROPERTY;
  const int wi = same ? 5 : 6;
  const float* tm = b.tmap + (size_t)nd.tslot * Mp;
  const float* pf = b.pfc + (size_t)nd.pslot * 2 * POOL_PARTS * Mp;
  const int tid = threadIdx.x;
  float lss = 0.f;
  for (int c = tid; c < Mp; c += MT) {
    float a0 = 0.f, a1 = 1.f, t = 0.f, v = 0.f;
    if (c < M) {
      a0 = w.batt[same ? 1 : 3][c];
      for (int p = 0; p < POOL_PARTS; ++p) a0 += pf[p * Mp + c];
      t = tm[c];
      v = a0 * t;
      if (same) {
        a1 = w.batt[2][c];
        for (int p = 0; p < POOL_PARTS; ++p) a1 += pf[(POOL_PARTS + p) * Mp + c];
        v *= a1;
      }
    }
    a0s[c] = a0; a1s[c] = a1; tms[c] = t; evs[c] = v;
    lss += v * v;
  }
  const bool big = g.hb_den != nullptr;
  if (!big)
    for (int c = tid; c < C; c += MT) {
      const float d = g.dscores[(size_t)nd.out_row * C + c];
      ds[c] = d;
      atomicAdd(g.gbans[wi] + c, d);
    }
  const float ss = block_reduce<0>(lss, scratch);      // (also orders the LDS writes above)
  const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  // den[m] = W_e[m,:] . ds ;  dW_e[m,c] += en[m] ds[c]
  float ldot = 0.f;
  float den_l[4];                 // Mp <= 1024 with 256 threads
  for (int c = tid, q = 0; c < Mp; c += MT, ++q) {
    float den = 0.f;
    if (big) {                    // den comes from the batch GEMM; keep en for the weight gradient
      if (c < M) {
        den = g.hb_den[(size_t)nd.out_row * Mp + c];
        ldot += evs[c] * den;
      }
      g.hb_en[(size_t)nd.out_row * Mp + c] = c < M ? evs[c] * inv : 0.f;
    } else if (c < M) {
      const float en = evs[c] * inv;
      const float* wr = w.Wans[wi] + (size_t)c * C;
      float* gw = g.gWans[wi] + (size_t)c * C;
      // loads first, atomics afterwards, in chunks: interleaved, every W_e load waits behind the
      // preceding atomic (they may alias as far as the compiler knows) -- C round trips in a chain
      for (int k0 = 0; k0 < C; k0 += 8) {
        float wv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv8[j] = wr[min(k0 + j, C - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + j < C) den += wv8[j] * ds[k0 + j];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + j < C) atomicAdd(gw + k0 + j, en * ds[k0 + j]);
      }
      ldot += evs[c] * den;
    }
    den_l[q] = den;
  }
  if (big && tid == 0) g.hb_sel[nd.out_row] = 1;
  const float xdy = block_reduce<0>(ldot, scratch);    // ev . den
  const float k3 = l2n_k(ss, inv);
  float* dt = g.dtmap + (size_t)nd.tslot * Mp;
  float* da0 = g.dpfc + ((size_t)nd.pslot * 2 + 0) * Mp;
  float* da1 = g.dpfc + ((size_t)nd.pslot * 2 + 1) * Mp;
  for (int c = tid, q = 0; c < Mp; c += MT, ++q) {
    float vt = 0.f, v0 = 0.f, v1 = 0.f;
    if (c < M) {
      const float dev = inv * den_l[q] - k3 * evs[c] * xdy;
      const float a0 = a0s[c], a1 = a1s[c], t = tms[c];   // a1 == 1 for Describe
      vt = dev * a0 * a1;
      v0 = dev * t * a1;
      v1 = dev * a0 * t;
    }
    dt[c] = vt;
    da0[c] = v0;
    if (same) da1[c] = v1;
  }
}

// ---------------------------------------------------------------------------------------------
// pool_bwd: a = softmax_HW(logits); pooled = sum_hw a[hw] feat[n,hw,:]; A = pooled . W_att + b
// given dA (dpfc):  dpooled = W_att . dA;  da[hw] = feat[n,hw,:] . dpooled  (the HBM-bound read
// of the [H*W, D] map, shared by both inputs of SameProperty);  dlogit = a * (da - sum a da).
//
// pool_bwd_kernel: one workgroup per (pooling node, channel part): the part's rows of W_att are read
// one row per wave (coalesced 1-KB rows, wave reduction) -- a thread streaming its own row, as the
// first version did, touches 64 cache lines per load instruction and made the kernel 40-58 us --
// then the part's slice of every feature row is dotted with dpooled and the partial da[hw] is added
// into gda (zeroed with the rest of the per-step block).  pool_bwd_fin_kernel: softmax backward of
// the summed da.  dW_att comes from gemm_tn over the saved pooled features.
// ---------------------------------------------------------------------------------------------
constexpr int PB_T = 512;
constexpr int PB_PARTS = 4;
__global__ __launch_bounds__(PB_T) void pool_bwd_kernel(ModuleWeights w, ModuleBuffers b,
                                                        ModuleGrads g, int tab_off, int stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int node_id = b.tab[tab_off + blockIdx.x * stride];
  const DevNode nd = b.nodes[node_id];
  const int HW = b.H * b.W, D = b.D, Mp = b.Mp;
  const int HWq = (HW + 3) & ~3;
  const int part = blockIdx.y;
  const int dper = ((D / PB_PARTS) + 3) & ~3;        // channels of this part (multiple of 4)
  const int d0 = part * dper, d1 = min(D, d0 + dper);
  float* dA = smem;                 // [2][Mp]
  float* dpl = dA + 2 * Mp;         // [2][dper]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = PB_T / 64;
  const int nin = nd.op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
  for (int i = tid; i < nin * Mp; i += PB_T)
    dA[i] = g.dpfc[((size_t)nd.pslot * 2 + i / Mp) * Mp + (i % Mp)];
  __syncthreads();
  // dpooled_i[d] = sum_m W_att[d][m] dA_i[m]: wave per row d, lanes over m (16-B pieces)
  for (int i = 0; i < nin; ++i) {
    int wi;
    if (nd.op == N2NMN_OP_FIND_SAME_PROPERTY) wi = 0;
    else if (nd.op == N2NMN_OP_SAME_PROPERTY) wi = 1 + i;
    else wi = 3;
    const float* Wm = w.Watt[wi];
    if (part == 0)
      for (int m = tid; m < b.M; m += PB_T) atomicAdd(g.gbatt[wi] + m, dA[i * Mp + m]);   // d b_att
    constexpr int U1 = 8;                       // rows of a wave in flight
    for (int dbase = d0 + wv; dbase < d1; dbase += U1 * NW) {
      float sacc[U1];
      const float* wrow[U1];
#pragma unroll
      for (int u = 0; u < U1; ++u) {
        sacc[u] = 0.f;
        wrow[u] = Wm + (size_t)min(dbase + u * NW, d1 - 1) * Mp;   // clamped: loads unconditional
      }
      for (int m = 4 * lane; m < Mp; m += 256) {       // all U1 row loads issued back to back
        const float4 a4 = *reinterpret_cast<const float4*>(dA + i * Mp + m);
        float4 w4[U1];
#pragma unroll
        for (int u = 0; u < U1; ++u) w4[u] = *reinterpret_cast<const float4*>(wrow[u] + m);
#pragma unroll
        for (int u = 0; u < U1; ++u)
          sacc[u] += w4[u].x * a4.x + w4[u].y * a4.y + w4[u].z * a4.z + w4[u].w * a4.w;
      }
#pragma unroll
      for (int u = 0; u < U1; ++u) {
        const int d = dbase + u * NW;
        const float t = wave_sum(sacc[u]);
        if (lane == 0 && d < d1) dpl[i * dper + (d - d0)] = t;
      }
    }
  }
  __syncthreads();
  // partial da_i[hw] = feat[n, hw, d0:d1] . dpooled_i[d0:d1]: half a wave per row, 4 rows of a
  // half-wave in flight
  {
    const float* fb = b.feat + (size_t)nd.n * HW * D + d0;
    const int hl = lane & 31, hsel = lane >> 5;
    const int dn = d1 - d0;
    float* gd = g.gda + (size_t)nd.pslot * 2 * HWq;
    constexpr int U3 = 10;                      // rows of a half-wave in flight (16 half-waves x 10 >= 150)
    for (int r0 = 2 * wv + hsel; r0 < HW; r0 += U3 * 2 * NW) {
      float s0[U3], s1[U3];
      const float* frow[U3];
#pragma unroll
      for (int u = 0; u < U3; ++u) {
        s0[u] = 0.f; s1[u] = 0.f;
        frow[u] = fb + (size_t)min(r0 + u * 2 * NW, HW - 1) * D;
      }
      for (int d = 4 * hl; d < dn; d += 128) {         // all U3 row loads issued back to back
        float4 f4[U3];
#pragma unroll
        for (int u = 0; u < U3; ++u) f4[u] = *reinterpret_cast<const float4*>(frow[u] + d);
        const float4 p0 = *reinterpret_cast<const float4*>(dpl + d);
#pragma unroll
        for (int u = 0; u < U3; ++u)
          s0[u] += f4[u].x * p0.x + f4[u].y * p0.y + f4[u].z * p0.z + f4[u].w * p0.w;
        if (nin == 2) {
          const float4 p1 = *reinterpret_cast<const float4*>(dpl + dper + d);
#pragma unroll
          for (int u = 0; u < U3; ++u)
            s1[u] += f4[u].x * p1.x + f4[u].y * p1.y + f4[u].z * p1.z + f4[u].w * p1.w;
        }
      }
#pragma unroll
      for (int u = 0; u < U3; ++u) {
        const int r = r0 + u * 2 * NW;
        float t0 = s0[u], t1 = s1[u];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {      // reduce inside each half-wave
          t0 += __shfl_xor(t0, off);
          if (nin == 2) t1 += __shfl_xor(t1, off);
        }
        if (hl == 0 && r < HW) {
          atomicAdd(gd + r, t0);
          if (nin == 2) atomicAdd(gd + HWq + r, t1);
        }
      }
    }
  }
}

// d logits of the pooling node's inputs from the summed da (one workgroup per node)
__global__ __launch_bounds__(256) void pool_bwd_fin_kernel(ModuleBuffers b, ModuleGrads g,
                                                           int tab_off, int stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int node_id = b.tab[tab_off + blockIdx.x * stride];
  const DevNode nd = b.nodes[node_id];
  const int HW = b.H * b.W, tid = threadIdx.x;
  const int HWq = (HW + 3) & ~3;
  float* scratch = smem;            // [16]
  float* as_ = smem + 16;           // [2][HWq]
  const int nin = nd.op == N2NMN_OP_SAME_PROPERTY ? 2 : 1;
  for (int i = 0; i < nin; ++i) {
    const float* src = b.arena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    const float* gd = g.gda + ((size_t)nd.pslot * 2 + i) * HWq;
    float lm = -INFINITY;
    for (int r = tid; r < HW; r += 256) lm = fmaxf(lm, src[r]);
    const float mx = block_reduce<1>(lm, scratch);
    float ls = 0.f;
    for (int r = tid; r < HW; r += 256) {
      const float ex = expf(src[r] - mx);      // softmax as in the forward pool_kernel
      as_[i * HWq + r] = ex;
      ls += ex;
    }
    const float sum = block_reduce<0>(ls, scratch);
    float la = 0.f;
    for (int r = tid; r < HW; r += 256) {
      const float a = as_[i * HWq + r] / sum;
      as_[i * HWq + r] = a;
      la += a * gd[r];
    }
    const float sad = block_reduce<0>(la, scratch);
    float* go = g.garena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    for (int r = tid; r < HW; r += 256) go[r] = as_[i * HWq + r] * (gd[r] - sad);
  }
}

// ---------------------------------------------------------------------------------------------
// Find-type epilogue backward on rows [r0, r1) of the hoisted conv_image map (forward:
// find_epilogue in kernels_modules.hip):   att[r] = l2n_c(M[r,c] tt[c]) . w_e + b_e,
// tt = tmap (Find/Filter) or tmap * amap (FindSameProperty); Filter = min(input_0, att).
// ---------------------------------------------------------------------------------------------
__device__ void find_epilogue_bwd(const ModuleWeights& w, const ModuleBuffers& b,
                                  const ModuleGrads& g, const DevNode& nd, int node_id, int part,
                                  int nparts, float* smem) {
  constexpr int MAXCI = 4;
  const int HW = b.H * b.W, Mp = b.Mp, M = b.M;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool fsp = nd.op == N2NMN_OP_FIND_SAME_PROPERTY;
  const int wsel = fsp ? 1 : 0;
  const float* Mbuf = (fsp ? b.mfsp : b.mfind) + (size_t)nd.mslot * HW * Mp;
  float* dMbuf = (fsp ? g.dmfsp : g.dmfind) + (size_t)nd.mslot * HW * Mp;
  const float* tm = b.tmap + (size_t)nd.tslot * Mp;
  const float be = w.be[wsel][0];
  const int rpp = (HW + nparts - 1) / nparts;
  const int r0 = part * rpp, r1 = min(HW, r0 + rpp);
  const float* in0 = (nd.op == N2NMN_OP_FILTER) ? b.arena + (size_t)nd.in0 * b.HWp : nullptr;
  float* gin0 = in0 ? g.garena + (size_t)nd.in0 * b.HWp : nullptr;
  const float* gout = g.garena + (size_t)node_id * b.HWp;
  float* red = smem;                      // [4 waves][2][Mp]
  float* scratch = red + 8 * Mp;          // [16]
  float* xp = scratch + 16 + wid * 256;   // [4 waves][256]: lane <-> channel transpose for dM

  float4 t4[MAXCI], e4[MAXCI];
  float4 dtt[MAXCI], dwe[MAXCI];
#pragma unroll
  for (int i = 0; i < MAXCI; ++i) {
    const int c = 4 * lane + 256 * i;
    dtt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dwe[i] = dtt[i];
    if (c < Mp) {
      t4[i] = *reinterpret_cast<const float4*>(tm + c);
      e4[i] = *reinterpret_cast<const float4*>(w.we[wsel] + c);
      if (fsp) {
        float4 a4 = *reinterpret_cast<const float4*>(w.batt[0] + c);
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
  float dbe = 0.f;
  // a wave takes rows r0 + wid, + 4, ...; the map rows (and gradients) of UR of them are fetched
  // before any is processed -- one row per trip left every trip waiting on its own round trip
  constexpr int UR = 5;
  constexpr int RSTEP = MT / 64;
  for (int rb = r0 + wid; rb < r1; rb += UR * RSTEP) {
    float4 mrow[UR][MAXCI];
    float grv[UR], xin[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int r = min(rb + u * RSTEP, r1 - 1);       // clamped: loads stay unconditional
#pragma unroll
      for (int i = 0; i < MAXCI; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < Mp) mrow[u][i] = *reinterpret_cast<const float4*>(Mbuf + (size_t)r * Mp + c);
      }
      grv[u] = gout[r];
      xin[u] = in0 ? in0[r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int r = rb + u * RSTEP;
      if (r >= r1) break;                               // wave-uniform
      float4* m4 = mrow[u];
      float ss = 0.f, dot = 0.f;
#pragma unroll
      for (int i = 0; i < MAXCI; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < Mp) {
          const float p0 = m4[i].x * t4[i].x, p1 = m4[i].y * t4[i].y, p2 = m4[i].z * t4[i].z,
                      p3 = m4[i].w * t4[i].w;
          ss += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
          dot += p0 * e4[i].x + p1 * e4[i].y + p2 * e4[i].z + p3 * e4[i].w;
        }
      }
      ss = wave_sum(ss);
      dot = wave_sum(dot);
      const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
      float gr = grv[u];
      if (in0) {                                     // Filter: tf.minimum(input_0, att), tie -> input_0
        // att recomputed with the forward's exact arithmetic (find_epilogue: d2 / sqrt(...) + be):
        // nested Filters with identical text parameters (length-1 questions) produce EXACT ties,
        // and a 1-ulp difference in the recomputation would flip the branch
        const float att = dot / sqrtf(fmaxf(ss, 1e-12f)) + be;
        const float x = xin[u];
        const bool to_x = x <= att;
        if (lane == 0) gin0[r] = to_x ? gr : 0.f;
        gr = to_x ? 0.f : gr;
      }
      dbe += gr;
      // dP[c] = gr * (inv * w_e[c] - k3 * dot * P[c]),   P = M * tt
      const float ka = gr * inv, kb = gr * l2n_k(ss, inv) * dot;
#pragma unroll
      for (int i = 0; i < MAXCI; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < Mp) {
          const float p0 = m4[i].x * t4[i].x, p1 = m4[i].y * t4[i].y, p2 = m4[i].z * t4[i].z,
                      p3 = m4[i].w * t4[i].w;
          const float d0 = ka * e4[i].x - kb * p0, d1 = ka * e4[i].y - kb * p1,
                      d2 = ka * e4[i].z - kb * p2, d3 = ka * e4[i].w - kb * p3;
          dwe[i].x += ka * p0; dwe[i].y += ka * p1; dwe[i].z += ka * p2; dwe[i].w += ka * p3;
          dtt[i].x += d0 * m4[i].x; dtt[i].y += d1 * m4[i].y; dtt[i].z += d2 * m4[i].z;
          dtt[i].w += d3 * m4[i].w;
          if (gr != 0.f) {                            // wave-uniform
            // several nodes may share a conv_image slab -> atomic adds; a lane holds 4 consecutive
            // channels, so adding them directly is four 16-B-strided atomics (8 cache lines each,
            // 30 of the kernel's 51 us): transposed through LDS each atomic covers 64 consecutive
            // channels (2 lines)
            *reinterpret_cast<float4*>(xp + 4 * lane) =
                make_float4(d0 * t4[i].x, d1 * t4[i].y, d2 * t4[i].z, d3 * t4[i].w);
            float* dm = dMbuf + (size_t)r * Mp + 256 * i;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int cc = lane + 64 * k;
              const float v = xp[cc];
              if (256 * i + cc < M) atomicAdd(dm + cc, v);
            }
          }
        }
      }
    }
  }
  // reduce the per-wave channel accumulators
#pragma unroll
  for (int i = 0; i < MAXCI; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < Mp) {
      *reinterpret_cast<float4*>(red + (size_t)(wid * 2 + 0) * Mp + c) = dtt[i];
      *reinterpret_cast<float4*>(red + (size_t)(wid * 2 + 1) * Mp + c) = dwe[i];
    }
  }
  const float dbe_t = block_reduce<0>(lane == 0 ? dbe : 0.f, scratch);   // also a barrier
  for (int c = tid; c < M; c += MT) {
    float st = 0.f, sw = 0.f;
#pragma unroll
    for (int q = 0; q < MT / 64; ++q) { st += red[(size_t)(q * 2) * Mp + c]; sw += red[(size_t)(q * 2 + 1) * Mp + c]; }
    atomicAdd(g.gwe[wsel] + c, sw);
    if (fsp) {
      // tt = tmap * amap: amap recomputed per channel
      float am = w.batt[0][c];
      const float* pf = b.pfc + (size_t)nd.pslot * 2 * POOL_PARTS * Mp + c;
      for (int p = 0; p < POOL_PARTS; ++p) am += pf[p * Mp];
      atomicAdd(g.dtmap + (size_t)nd.tslot * Mp + c, st * am);
      atomicAdd(g.dpfc + ((size_t)nd.pslot * 2) * Mp + c, st * tm[c]);
    } else {
      atomicAdd(g.dtmap + (size_t)nd.tslot * Mp + c, st);
    }
  }
  if (tid == 0) atomicAdd(g.gbe[wsel], dbe_t);
}

// ---------------------------------------------------------------------------------------------
// Transform backward (forward: transform_op).  conv[p,c] = sum_tap K[tap,c] x[p+tap] + bt[c];
// v = conv * tm[c];  att[p] = l2n_c(v) . w_e + b_e.  Like the forward, a node is split over
// `nparts` workgroups by output pixels (<= 64 each); the parts are independent:
//   pass A (lanes = own pixels, waves = channel quarters): per-pixel ss, dot -> coefficients
//       dv[p,c] = A_p w_e[c] - B_p v[p,c];  d input[p + tap] += sum_c dv tm[c] K[tap,c]
//       (atomic into garena[input], which is zeroed at the start of the backward pass)
//   pass B (threads = channels, loop over own pixels): dK[tap,c], dbt[c], dtm[c], dw_e[c] partials,
//       added atomically (dtmap / the flat gradient buffer are zeroed as well)
// ---------------------------------------------------------------------------------------------
template <int KS>
__device__ void transform_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                              const DevNode& nd, int node_id, int part, int nparts, float* smem) {
  constexpr int KK = KS * KS;
  constexpr int RS = (KK + 3 + 3) & ~3;      // taps + bt + we + tm
  constexpr int PAD = KS / 2;
  const int H = b.H, W = b.W, HW = H * W, M = b.M, Mp = b.Mp;
  const int PW = W + 2 * PAD, PH = H + 2 * PAD;
  float* Kl = smem;                            // [M][RS]: K[tap] (raw), bt, we, tm
  float* xin = Kl + (size_t)M * RS;            // [PH][PW]
  float* Ap = xin + ((PH * PW + 3) & ~3);      // [64]
  float* Bp = Ap + 64;                         // [64]
  float* red = Bp + 64;                        // [4 waves][64][2]
  float* scratch = red + 4 * 64 * 2;           // [16]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* in0 = b.arena + (size_t)nd.in0 * b.HWp;
  const float* tm = b.tmap + (size_t)nd.tslot * Mp;
  const float* gout = g.garena + (size_t)node_id * b.HWp;
  const int ppp = (HW + nparts - 1) / nparts;
  const int p0 = part * ppp, p1 = min(HW, p0 + ppp);
  if (p1 - p0 > 64) return;                    // (forward guarantees <= 64 pixels per part)
  for (int i = tid; i < PH * PW; i += MT) {
    const int y = i / PW - PAD, x = i % PW - PAD;
    xin[i] = (y >= 0 && y < H && x >= 0 && x < W) ? in0[y * W + x] : 0.f;
  }
  for (int i = tid; i < KK * M; i += MT) {
    const int tap = i / M, c = i - tap * M;
    Kl[c * RS + tap] = w.Kt[i];
  }
  for (int c = tid; c < M; c += MT) {
    Kl[c * RS + KK] = w.bt[c];
    Kl[c * RS + KK + 1] = w.we[2][c];
    Kl[c * RS + KK + 2] = tm[c];
  }
  __syncthreads();
  // ---- pass A: lanes = own pixels ----
  const int p = p0 + lane;
  const bool on = p < p1;
  const int py = on ? p / W : 0, px = on ? p - (p / W) * W : 0;
  float win[KK];
#pragma unroll
  for (int dy = 0; dy < KS; ++dy)
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) win[dy * KS + dx] = xin[(py + dy) * PW + px + dx];
  float ss = 0.f, dot = 0.f;
  for (int c = wid; c < M; c += MT / 64) {
    const float4* kr = reinterpret_cast<const float4*>(Kl + (size_t)c * RS);
    float kk[RS];
#pragma unroll
    for (int q = 0; q < RS / 4; ++q) {
      const float4 t4 = kr[q];
      kk[4 * q] = t4.x; kk[4 * q + 1] = t4.y; kk[4 * q + 2] = t4.z; kk[4 * q + 3] = t4.w;
    }
    float cv = kk[KK];
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) cv += kk[tap] * win[tap];
    const float v = cv * kk[KK + 2];
    ss += v * v;
    dot += v * kk[KK + 1];
  }
  red[(wid * 64 + lane) * 2] = ss;
  red[(wid * 64 + lane) * 2 + 1] = dot;
  __syncthreads();
  float s2 = 0.f, d2 = 0.f;
#pragma unroll
  for (int q = 0; q < MT / 64; ++q) { s2 += red[(q * 64 + lane) * 2]; d2 += red[(q * 64 + lane) * 2 + 1]; }
  const float inv = 1.0f / sqrtf(fmaxf(s2, 1e-12f));
  const float gp = on ? gout[p] : 0.f;
  const float A = gp * inv, Bc = gp * l2n_k(s2, inv) * d2;
  if (wid == 0) { Ap[lane] = A; Bp[lane] = Bc; }
  float ga[KK];
#pragma unroll
  for (int tap = 0; tap < KK; ++tap) ga[tap] = 0.f;
  for (int c = wid; c < M; c += MT / 64) {
    const float4* kr = reinterpret_cast<const float4*>(Kl + (size_t)c * RS);
    float kk[RS];
#pragma unroll
    for (int q = 0; q < RS / 4; ++q) {
      const float4 t4 = kr[q];
      kk[4 * q] = t4.x; kk[4 * q + 1] = t4.y; kk[4 * q + 2] = t4.z; kk[4 * q + 3] = t4.w;
    }
    float cv = kk[KK];
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) cv += kk[tap] * win[tap];
    const float tmc = kk[KK + 2];
    const float dconv = (A * kk[KK + 1] - Bc * cv * tmc) * tmc;
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) ga[tap] += dconv * kk[tap];
  }
  if (on) {     // conv is a cross-correlation: input pixel = p + tap - PAD
    float* gin = g.garena + (size_t)nd.in0 * b.HWp;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const int y = py + dy - PAD, x = px + dx - PAD;
        if (y >= 0 && y < H && x >= 0 && x < W) atomicAdd(gin + y * W + x, ga[dy * KS + dx]);
      }
  }
  __syncthreads();                              // Ap / Bp visible
  // ---- pass B: threads = channels, own pixels ----
  for (int c = tid; c < M; c += MT) {
    const float* kr = Kl + (size_t)c * RS;
    float kreg[KK];
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) kreg[tap] = kr[tap];
    const float btc = kr[KK], wec = kr[KK + 1], tmc = kr[KK + 2];
    float dK[KK];
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) dK[tap] = 0.f;
    float dbt = 0.f, dtm = 0.f, dwe = 0.f;
    for (int q = p0; q < p1; ++q) {
      const int y = q / W, x = q - (q / W) * W;
      float cv = btc;
      float xw[KK];
#pragma unroll
      for (int dy = 0; dy < KS; ++dy)
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
          xw[dy * KS + dx] = xin[(y + dy) * PW + x + dx];       // broadcast read
          cv += kreg[dy * KS + dx] * xw[dy * KS + dx];
        }
      const float Aq = Ap[q - p0], Bq = Bp[q - p0];
      const float v = cv * tmc;
      const float dv = Aq * wec - Bq * v;
      dwe += Aq * v;                      // g_p * inv_p * v
      dtm += dv * cv;
      const float dconv = dv * tmc;
      dbt += dconv;
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) dK[tap] += dconv * xw[tap];
    }
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) atomicAdd(g.gKt + (size_t)tap * M + c, dK[tap]);
    atomicAdd(g.gbt + c, dbt);
    atomicAdd(g.gwe[2] + c, dwe);
    atomicAdd(g.dtmap + (size_t)nd.tslot * Mp + c, dtm);
  }
  const float gs = block_reduce<0>(wid == 0 ? gp : 0.f, scratch);
  if (tid == 0) atomicAdd(g.gbe[2], gs);
}

// ---------------------------------------------------------------------------------------------
// answer heads on raw attention maps (forward: light_answer): scores = x . W + b with
// Exist x = [min, mean, max]; Count x = [map, min, max]; *Num x = [map0, min0, max0, map1, min1, max1]
// ---------------------------------------------------------------------------------------------
__device__ void light_answer_bwd(const ModuleWeights& w, const ModuleBuffers& b,
                                 const ModuleGrads& g, const DevNode& nd, float* smem) {
  const int HW = b.H * b.W, C = b.C;
  float* x = smem;                              // [2*HW + 4]
  float* dx = x + ((2 * HW + 4 + 3) & ~3);      // [2*HW + 4]
  float* ds = dx + ((2 * HW + 4 + 3) & ~3);     // [C padded]
  float* scratch = ds + ((C + 31) & ~31);       // [16]
  const int nin = (nd.op == N2NMN_OP_EXIST || nd.op == N2NMN_OP_COUNT) ? 1 : 2;
  const int tid = threadIdx.x;
  float mn[2], mx[2], cmn[2], cmx[2];
  for (int i = 0; i < nin; ++i) {
    const float* src = b.arena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    float lmn = INFINITY, lmx = -INFINITY;
    for (int r = tid; r < HW; r += MT) {
      const float v = src[r];
      x[i * (HW + 2) + r] = v;
      lmn = fminf(lmn, v); lmx = fmaxf(lmx, v);
    }
    mn[i] = block_reduce<2>(lmn, scratch);
    mx[i] = block_reduce<1>(lmx, scratch);
    float c0 = 0.f, c1 = 0.f;
    for (int r = tid; r < HW; r += MT) {
      const float v = src[r];
      c0 += v == mn[i] ? 1.f : 0.f;
      c1 += v == mx[i] ? 1.f : 0.f;
    }
    cmn[i] = block_reduce<0>(c0, scratch);
    cmx[i] = block_reduce<0>(c1, scratch);
  }
  float sm0 = 0.f;
  if (nd.op == N2NMN_OP_EXIST) {
    float ls = 0.f;
    for (int r = tid; r < HW; r += MT) ls += x[r];
    sm0 = block_reduce<0>(ls, scratch);
  }
  __syncthreads();
  int F, wi;
  if (nd.op == N2NMN_OP_EXIST) {
    if (tid == 0) { x[0] = mn[0]; x[1] = sm0 / (float)HW; x[2] = mx[0]; }
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
  for (int c = tid; c < C; c += MT) {
    const float d = g.dscores[(size_t)nd.out_row * C + c];
    ds[c] = d;
    atomicAdd(g.gbans[wi] + c, d);
  }
  __syncthreads();
  for (int f = tid; f < F; f += MT) {
    const float* wr = w.Wans[wi] + (size_t)f * C;
    float* gw = g.gWans[wi] + (size_t)f * C;
    const float xf = x[f];
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      s += wr[c] * ds[c];
      atomicAdd(gw + c, xf * ds[c]);
    }
    dx[f] = s;
  }
  __syncthreads();
  for (int i = 0; i < nin; ++i) {
    const float* src = b.arena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    float* go = g.garena + (size_t)(i == 0 ? nd.in0 : nd.in1) * b.HWp;
    float dmn, dmx, dmean = 0.f;
    const int base = i * (HW + 2);
    if (nd.op == N2NMN_OP_EXIST) { dmn = dx[0]; dmean = dx[1] / (float)HW; dmx = dx[2]; }
    else { dmn = dx[base + HW]; dmx = dx[base + HW + 1]; }
    for (int r = tid; r < HW; r += MT) {
      const float v = src[r];
      float gr = nd.op == N2NMN_OP_EXIST ? dmean : dx[base + r];
      if (v == mn[i]) gr += dmn / cmn[i];
      if (v == mx[i]) gr += dmx / cmx[i];
      go[r] = gr;
    }
  }
}

__global__ __launch_bounds__(MT) void att_bwd_kernel(ModuleWeights w, ModuleBuffers b,
                                                     ModuleGrads g, int tab_off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int* e = b.tab + tab_off + blockIdx.x * 4;
  const int node_id = e[0], part = e[1], nparts = e[2];
  const DevNode nd = b.nodes[node_id];
  const int HW = b.H * b.W;
  switch (nd.op) {
    case N2NMN_OP_SCENE:
      break;
    case N2NMN_OP_FIND:
    case N2NMN_OP_FILTER:
    case N2NMN_OP_FIND_SAME_PROPERTY:
      find_epilogue_bwd(w, b, g, nd, node_id, part, nparts, smem);
      break;
    case N2NMN_OP_TRANSFORM:
      if (b.ksize == 5) transform_bwd<5>(w, b, g, nd, node_id, part, nparts, smem);
      else transform_bwd<3>(w, b, g, nd, node_id, part, nparts, smem);
      break;
    case N2NMN_OP_AND:
    case N2NMN_OP_OR: {
      const float* a0 = b.arena + (size_t)nd.in0 * b.HWp;
      const float* a1 = b.arena + (size_t)nd.in1 * b.HWp;
      const float* go = g.garena + (size_t)node_id * b.HWp;
      float* g0 = g.garena + (size_t)nd.in0 * b.HWp;
      float* g1 = g.garena + (size_t)nd.in1 * b.HWp;
      for (int r = threadIdx.x; r < HW; r += MT) {
        const bool first = nd.op == N2NMN_OP_AND ? a0[r] <= a1[r] : a0[r] >= a1[r];
        g0[r] = first ? go[r] : 0.f;
        g1[r] = first ? 0.f : go[r];
      }
      break;
    }
    default:
      light_answer_bwd(w, b, g, nd, smem);
      break;
  }
}

// ---------------------------------------------------------------------------------------------
// textmap_bwd: tmap[tslot,:] = word_vecs[t*N+n,:] . W_txt + b  ->  dword_vecs = dtmap . W_txt^T.
// Same work table as the forward (groups of <= TM_GROUP nodes sharing a weight set); a wave owns
// an embedding row e and reduces over the map channels.  dW_txt / db_txt come from gemm_tn/colsum.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MT) void textmap_bwd_kernel(ModuleWeights w, ModuleBuffers b,
                                                         ModuleGrads g, int tab_off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int* tab = b.tab + tab_off + blockIdx.x * (2 + TM_GROUP);
  const int ws = tab[0], cnt = tab[1];
  const int E = b.E, Mp = b.Mp;
  float* dt = smem;                         // [TM_GROUP][Mp]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // a wave copies whole d tmap rows: the node -> slot lookup (two dependent loads) once per row,
  // not once per element
  for (int gi = wid; gi < TM_GROUP; gi += MT / 64) {
    const int slot = gi < cnt ? b.nodes[tab[2 + gi]].tslot : -1;
    for (int c = 4 * lane; c < Mp; c += 256)
      *reinterpret_cast<float4*>(dt + gi * Mp + c) =
          slot >= 0 ? *reinterpret_cast<const float4*>(g.dtmap + (size_t)slot * Mp + c)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const float* Wp = w.Wtxt[ws];
  const int eper = (E + gridDim.y - 1) / gridDim.y;
  const int e0 = blockIdx.y * eper, e1 = min(E, e0 + eper);
  // a wave takes rows e0 + wid, + 4, ...: the W_txt rows of UE of them are fetched together
  constexpr int UE = 8;
  constexpr int MAXM4 = 4;                   // Mp <= 1024
  for (int eb = e0 + wid; eb < e1; eb += UE * (MT / 64)) {
    float4 w4[UE][MAXM4];
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const int e = min(eb + u * (MT / 64), e1 - 1);   // clamped: loads stay unconditional
#pragma unroll
      for (int i = 0; i < MAXM4; ++i) {
        const int m = 4 * lane + 256 * i;
        if (m < Mp) w4[u][i] = *reinterpret_cast<const float4*>(Wp + (size_t)e * Mp + m);
      }
    }
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const int e = eb + u * (MT / 64);
      if (e >= e1) break;                              // wave-uniform
      float s[TM_GROUP];
#pragma unroll
      for (int gi = 0; gi < TM_GROUP; ++gi) s[gi] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXM4; ++i) {
        const int m = 4 * lane + 256 * i;
        if (m < Mp) {
#pragma unroll
          for (int gi = 0; gi < TM_GROUP; ++gi) {
            const float4 d4 = *reinterpret_cast<const float4*>(dt + gi * Mp + m);
            s[gi] += w4[u][i].x * d4.x + w4[u][i].y * d4.y + w4[u][i].z * d4.z + w4[u][i].w * d4.w;
          }
        }
      }
#pragma unroll
      for (int gi = 0; gi < TM_GROUP; ++gi) {
        const float r = wave_sum(s[gi]);
        if (lane == 0 && gi < cnt) {
          const DevNode& nd = b.nodes[tab[2 + gi]];
          g.dwv[((size_t)nd.t * b.N_full + nd.n) * E + e] = r;
        }
      }
    }
  }
}

}  // namespace

void launch_heads_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                      int tab_off, int count, hipStream_t s) {
  const size_t smem = sizeof(float) * (4 * (size_t)b.Mp + ((b.C + 31) & ~31) + 16);
  hipLaunchKernelGGL(heads_bwd_kernel, dim3(count), dim3(MT), smem, s, w, b, g, tab_off);
}

void launch_pool_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                     int tab_off, int count, int stride, hipStream_t s) {
  const int dper = ((b.D / PB_PARTS) + 3) & ~3;
  const size_t smem = sizeof(float) * (2 * (size_t)b.Mp + 2 * (size_t)dper);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(count, PB_PARTS), dim3(PB_T), smem, s, w, b, g, tab_off,
                     stride);
  const int HWq = (b.H * b.W + 3) & ~3;
  hipLaunchKernelGGL(pool_bwd_fin_kernel, dim3(count), dim3(256),
                     sizeof(float) * (16 + 2 * (size_t)HWq), s, b, g, tab_off, stride);
}

void launch_att_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                    int tab_off, int count, hipStream_t s) {
  const int HW = b.H * b.W;
  const int pad = b.ksize / 2;
  const int KK = b.ksize * b.ksize;
  const int RS = (KK + 3 + 3) & ~3;
  const int HWq = (HW + 3) & ~3;
  const size_t tr = (size_t)b.M * RS + (((size_t)(b.H + 2 * pad) * (b.W + 2 * pad) + 3) & ~3) +
                    2 * 64 + 4 * 64 * 2 + 16;
  (void)HWq;
  const size_t fe = 8 * (size_t)b.Mp + 16 + 4 * 256;
  const size_t la = 2 * (size_t)((2 * HW + 4 + 3) & ~3) + ((b.C + 31) & ~31) + 16;
  const size_t smem = sizeof(float) * std::max(tr, std::max(fe, la));
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(att_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(att_bwd_kernel, dim3(count), dim3(MT), smem, s, w, b, g, tab_off);
}

void launch_textmap_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                        int tab_off, int count, hipStream_t s) {
  const size_t smem = sizeof(float) * ((size_t)TM_GROUP * b.Mp);
  // grid.y splits the embedding rows so that the ~N/8 node groups still fill the chip
  hipLaunchKernelGGL(textmap_bwd_kernel, dim3(count, 10), dim3(MT), smem, s, w, b, g, tab_off);
}

}  // namespace n2nmn
